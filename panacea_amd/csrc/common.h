// common.h — shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 f16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "panacea_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define PNC_WAVE 64

// MFMA C/D fragment map of v_mfma_f32_32x32x16_f16 (MI355X_MICROARCH / CDNA4 guide §3):
//   column = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
__device__ __forceinline__ int mfma32_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// 128-byte LDS rows (64 fp16) hold eight 16-byte chunks; XOR-swizzle the chunk index with
// (row>>1)&7 so that the 16-lane groups of ds_read_b128 (rows differ, chunk equal) hit
// sixteen distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ int lds_off128(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions): the results are rounded to fp16
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// LDS DMA (global_load_lds_dwordx4): lane l's 16 bytes land at lds_wave_base + 16*l — the LDS image of one wave
// instruction is lane-linear (1 KB), so any swizzle has to be applied to the SOURCE address
__device__ __forceinline__ void glds16(const half_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same DMA through a buffer resource: address = rsrc base (SGPRs) + per-lane 32-bit byte offset + scalar byte offset.  The
// per-lane offset of a tile row does not change along K (only the scalar offset does), and offsets at or beyond num_records
// read as zero — padding / tails need no zero-block pointer.
using buffer_rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void glds16_buf(buffer_rsrc_t rsrc, unsigned lane_byte_off, unsigned scalar_byte_off, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)lane_byte_off,
                                             (int)scalar_byte_off, 0, 0);
}
constexpr unsigned PNC_BUF_OOB = 0x80000000u;      // a lane offset no tensor of the path reaches

// Precise ("split") fp16 operands (include/panacea_hip.h, PncGemmParams.A_lo): v ~ hi + lo * 2^-11 with
// hi = fp16(v), lo = fp16((v - hi) * 2^11).
constexpr float PNC_LO_SCALE = 2048.0f;
__device__ __forceinline__ half4v lo_plane4(const float (&v)[4], half4v hi) {
    half4v l;
#pragma unroll
    for (int e = 0; e < 4; ++e) l[e] = (half_t)((v[e] - (float)hi[e]) * PNC_LO_SCALE);
    return l;
}

// RANGE MONITOR (round 6; include/panacea_hip.h pnc_range_monitor_collect).  The numeric contract of a split operand holds while its
// e4m3 lo plane does not saturate: lo = (v - fp16(v)) * 2^11 stays inside +-448 for |v| < 512 and clamps from there on (fp16's ulp
// reaches 0.5).  Every e4m3 lo plane of the library is packed by pack4_e4m3() below, so ONE counter per translation unit, bumped
// when a packed quad clamps (a compare on values the pack already holds; the atomic is never reached inside the range), states at
// run time whether an evaluation left the range its contract is written for.
static __device__ __attribute__((unused)) unsigned int pnc_tu_lo_clamped;
static __global__ __attribute__((unused)) void pnc_tu_collect_kernel(unsigned int* out) {
    const unsigned v = atomicExch(&pnc_tu_lo_clamped, 0u);
    if (v) atomicAdd(out, v);
}
// one host-side accessor per translation unit that packs lo planes; misc.hip's pnc_range_monitor_collect() calls them all
#define PNC_DEFINE_TU_COLLECT(name) \
    void pnc_tu_collect_##name(unsigned int* out, hipStream_t st) { hipLaunchKernelGGL(pnc_tu_collect_kernel, dim3(1), dim3(1), 0, st, out); }

// The same lo plane as OCP fp8 e4m3 (PNC_LO_E4M3): |r| <= |v|, clamped to the format's +-448; four consecutive channels -> one dword.
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    if (__builtin_expect(fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))) > 448.0f, 0)) atomicAdd(&pnc_tu_lo_clamped, 1u);
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ unsigned lo_plane4_e4m3(const float (&v)[4], half4v hi) {
    return pack4_e4m3((v[0] - (float)hi[0]) * PNC_LO_SCALE, (v[1] - (float)hi[1]) * PNC_LO_SCALE,
                      (v[2] - (float)hi[2]) * PNC_LO_SCALE, (v[3] - (float)hi[3]) * PNC_LO_SCALE);
}
// lo plane of 4 consecutive channels at ELEMENT index idx of the plane `lo` in either format (fmt is launch-uniform)
__device__ __forceinline__ void store_lo4(void* lo, int fmt, int64_t idx, const float (&v)[4], half4v hi) {
    if (fmt == PNC_LO_E4M3) *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lo) + idx) = lo_plane4_e4m3(v, hi);
    else *reinterpret_cast<half4v*>(reinterpret_cast<half_t*>(lo) + idx) = lo_plane4(v, hi);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block id -> XCD-contiguous tile id (bijective for any nblk; guide §5 "XCD swizzle must be bijective")
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

int pnc_get_option(int option);          // misc.hip (process-global tuning / test switches, see pnc_set_option)

static inline int pnc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PNC_OK : (int)e;
}
