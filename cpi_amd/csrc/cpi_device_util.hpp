// cpi_device_util.hpp -- device helpers shared by all kernels: loads / stores, DPP moves, wave reductions, kernel argument blocks.
// Included by every kernel translation unit after cpi_args.hpp / cpi_math.hpp (not a stand-alone header).
#pragma once

namespace {

// ============================================================================================
// device helpers
// ============================================================================================

__device__ __forceinline__ V3 ldv3(const double *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ Q4 ldq4(const double *p) { Q4 q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }
__device__ __forceinline__ M3 ldm3_cm(const double *p) {  // column-major 3x3
    M3 A;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) A.m[i][j] = p[j * 3 + i];
    return A;
}
__device__ __forceinline__ void stm3_cm(double *p, const M3 &A) {
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) p[j * 3 + i] = A.m[i][j];
}
__device__ __forceinline__ void stv3(double *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

__device__ __forceinline__ V3 shfl_down(V3 v, int d) {
    return mk(__shfl_down(v.x, d), __shfl_down(v.y, d), __shfl_down(v.z, d));
}
__device__ __forceinline__ M3 shfl_down(const M3 &A, int d) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = __shfl_down(A.m[i][j], d);
    return r;
}
__device__ __forceinline__ V3 shfl_down(V3 v, int d, int width) {
    return mk(__shfl_down(v.x, d, width), __shfl_down(v.y, d, width), __shfl_down(v.z, d, width));
}
__device__ __forceinline__ M3 shfl_up(const M3 &A, int d, int width) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = __shfl_up(A.m[i][j], d, width);
    return r;
}
// Shifts inside a 16-lane DPP row (= one model-1 window group of the covariance kernel): v_mov_b32_dpp row_shr / row_shl
// instead of ds_bpermute -- VALU moves with no LDS round trip to wait for.  Lanes whose source would lie outside the row
// keep their own value, exactly like __shfl_up / __shfl_down with width 16.  d is a constant after unrolling.
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double row16_up(double v, int d) {      // lane j <- lane j - d
    switch (d) {
        case 1: return dpp_mov64<0x111>(v);
        case 2: return dpp_mov64<0x112>(v);
        case 4: return dpp_mov64<0x114>(v);
        case 8: return dpp_mov64<0x118>(v);
        default: return __shfl_up(v, d, 16);
    }
}
__device__ __forceinline__ double row16_down(double v, int d) {    // lane j <- lane j + d
    switch (d) {
        case 1: return dpp_mov64<0x101>(v);
        case 2: return dpp_mov64<0x102>(v);
        case 4: return dpp_mov64<0x104>(v);
        case 8: return dpp_mov64<0x108>(v);
        default: return __shfl_down(v, d, 16);
    }
}
__device__ __forceinline__ M3 row16_up_m3(const M3 &A, int d) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = row16_up(A.m[i][j], d);
    return r;
}
// Lane 15 of DPP rows 0 and 2 -> every lane of rows 1 and 3 (row_bcast:15, row_mask 0b1010); rows 0 and 2 keep their value.
__device__ __forceinline__ M3 row_bcast15_to_odd_rows(const M3 &A) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int lo = __double2loint(A.m[i][j]), hi = __double2hiint(A.m[i][j]);
            r.m[i][j] = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xa, 0xf, false),
                                         __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xa, 0xf, false));
        }
    return r;
}
template <int GROUP>
__device__ __forceinline__ M3 group_up(const M3 &A, int d) {
    if (GROUP != 16) return shfl_up(A, d, GROUP);
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = row16_up(A.m[i][j], d);
    return r;
}
template <int GROUP>
__device__ __forceinline__ V3 group_down(V3 v, int d) {
    if (GROUP != 16) return shfl_down(v, d, GROUP);
    return mk(row16_down(v.x, d), row16_down(v.y, d), row16_down(v.z, d));
}
template <int GROUP>
__device__ __forceinline__ double group_down(double v, int d) {
    return (GROUP != 16) ? __shfl_down(v, d, GROUP) : row16_down(v, d);
}
template <bool JAC>
__device__ __forceinline__ MeanState<JAC> shfl_down(const MeanState<JAC> &s, int d) {
    MeanState<JAC> r;
    r.R = shfl_down(s.R, d);
    r.alpha = shfl_down(s.alpha, d);
    r.beta = shfl_down(s.beta, d);
    r.DT = __shfl_down(s.DT, d);
    if (JAC) {
        r.Jq = shfl_down(s.Jq, d); r.Ja = shfl_down(s.Ja, d); r.Jb = shfl_down(s.Jb, d);
        r.Ha = shfl_down(s.Ha, d); r.Hb = shfl_down(s.Hb, d);
        r.Oa = s.Oa; r.Ob = s.Ob;
    }
    return r;
}
__device__ __forceinline__ GravAcc shfl_down(const GravAcc &g, int d) {
    GravAcc r;
    r.Gam = shfl_down(g.Gam, d); r.Lam = shfl_down(g.Lam, d);
    return r;
}
// Orders LDS traffic of a single-wavefront workgroup for the COMPILER only: a wave's DS instructions execute
// in issue order, so no counter drain (and no s_barrier) is needed between a write and a dependent read.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// lanes 4..7 of DPP rows 0 and 2 (= the clone lanes of the two 32-lane model-2 groups) <- lanes 0..3 of the same
// row; every other lane keeps its value.  row_shr:4, row_mask 0b0101, bank_mask 0b0010, bound_ctrl 0.
__device__ __forceinline__ double dpp_clone_shr4(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int nlo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0x5, 0x2, false);
    const int nhi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0x5, 0x2, false);
    return __hiloint2double(nhi, nlo);
}
// lanes 12..15 of every DPP row <- lanes 6..9 of the same row (row_shr:6, bank_mask 0b1000); all other lanes keep `old`.
__device__ __forceinline__ double dpp_shr6_bank3(double old, double src) {
    const int nlo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x116, 0xf, 0x8, false);
    const int nhi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x116, 0xf, 0x8, false);
    return __hiloint2double(nhi, nlo);
}
// the same for the odd DPP rows only (row_mask 0b1010): lanes 28..31 / 60..63 <- 22..25 / 54..57 -- model 2's p lanes take
// the stage value of its v lanes (cpi_math.hpp: CPI_COV2_PSYM lane map)
__device__ __forceinline__ double dpp_shr6_bank3_oddrows(double old, double src) {
    const int nlo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x116, 0xa, 0x8, false);
    const int nhi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x116, 0xa, 0x8, false);
    return __hiloint2double(nhi, nlo);
}
// Maximum over the wavefront, wave-uniform result.  DPP reduction (row_shr 1/2/4/8 -> lane 15 of each row holds the row
// maximum; row_bcast:15 / row_bcast:31 carry it across rows; lane 63 holds the total) instead of six dependent
// ds_bpermute round trips: it sits on the critical path of every wavefront's start-up.
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// LDS-DMA: 16 bytes per lane from global memory straight into LDS, no vector registers in between.  The LDS image of one
// instruction is fixed by the hardware -- "wave-uniform base (M0) + 16 x lane" --, the global address is "scalar base +
// 32-bit lane offset"; gfx950 takes 8-byte aligned sources (56-byte knots).  hipcc does not preserve M0 around a statement, so
// it is set and restored here.  Ordering is the wave's own counted s_waitcnt vmcnt (the compiler's counter model does not see
// these loads: wait explicitly with wait_vmcnt); a region is re-armed only after lgkmcnt(0) has retired the reads of its
// previous contents.
__device__ __forceinline__ void glds16(unsigned voff, const void *sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); }
__device__ __forceinline__ long long readfirstlane64(long long v) {
    return ((long long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
}

// (PreArgs: cpi_args.hpp)


}  // namespace
