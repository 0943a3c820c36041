// ubench_valu2.hip -- second table: issue cost of the remaining opcodes the sweeps use (or could use instead).
// Same method as ubench_valu.hip, one row per opcode at k = 2 and 4 waves per SIMD (the first table showed that a
// single wave issues one VALU per ~5 cycles whatever the opcode, and that the SIMD rate is reached from 2 waves on).
// Each row: cycles per wave-instruction per SIMD = median wave cycles / (instructions * k).
//
//   hipcc --offload-arch=gfx950 -O2 -o ubench_valu2 tools/ubench_valu2.hip && ./ubench_valu2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLK8(STMT) STMT STMT STMT STMT STMT STMT STMT STMT
#define ACC8F "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define ACC8U "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)

// OPS(X): X(id, "label", body) -- body issues exactly 64 instructions of the opcode per trip
#define T2F(NAME, INS) X(NAME, INS " d, d, b", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
    INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
    : ACC8F : "v"(b)); })
#define T2U(NAME, INS) X(NAME, INS " d, d, b", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
    INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
    : ACC8U : "v"(ub)); })
#define T3F(NAME, INS) X(NAME, INS " d, d, b, c", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
    INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
    : ACC8F : "v"(b), "v"(c)); })
#define T3U(NAME, INS) X(NAME, INS " d, d, b, c", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
    INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
    : ACC8U : "v"(ub), "v"(uc)); })
#define T1F(NAME, INS) X(NAME, INS " d, d", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
    INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n" : ACC8F); })
#define T1U(NAME, INS) X(NAME, INS " d, d", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
    INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n" : ACC8U); })

#define OPS \
    T3F(FMA, "v_fma_f32") \
    T2F(FMAC, "v_fmac_f32") \
    T2F(SUB, "v_sub_f32") \
    T2F(MIN, "v_min_f32") \
    T2F(MAX, "v_max_f32") \
    T3F(MED3, "v_med3_f32") \
    T2F(PKMUL_AS_1, "v_mul_legacy_f32") \
    T1F(SQRT, "v_sqrt_f32") \
    T1F(CVT_I2F, "v_cvt_f32_i32") \
    T1F(FRACT, "v_fract_f32") \
    T1F(FLOOR, "v_floor_f32") \
    T2U(ADD_U32, "v_add_u32") \
    T2U(OR, "v_or_b32") \
    T2U(XOR, "v_xor_b32") \
    T2U(LSHLREV, "v_lshlrev_b32") \
    T2U(LSHRREV, "v_lshrrev_b32") \
    T2U(MUL_LO, "v_mul_lo_u32") \
    T2U(MUL_U24, "v_mul_u32_u24") \
    T3U(MAD_U24, "v_mad_u32_u24") \
    T3U(LSHL_ADD, "v_lshl_add_u32") \
    T3U(LSHL_OR, "v_lshl_or_b32") \
    T3U(AND_OR, "v_and_or_b32") \
    T3U(ADD3, "v_add3_u32") \
    T3U(BFE, "v_bfe_u32") \
    T3U(BFI, "v_bfi_b32") \
    T3U(ALIGNBIT, "v_alignbit_b32") \
    T3U(PERM, "v_perm_b32") \
    T1U(BCNT0, "v_ffbh_u32") \
    T1U(FFBL, "v_ffbl_b32") \
    T1U(BFREV, "v_bfrev_b32") \
    T1U(NOT, "v_not_b32") \
    X(BCNT, "v_bcnt_u32_b32 d, d, b", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_bcnt_u32_b32 %0, %0, %8\nv_bcnt_u32_b32 %1, %1, %8\nv_bcnt_u32_b32 %2, %2, %8\nv_bcnt_u32_b32 %3, %3, %8\n" \
        "v_bcnt_u32_b32 %4, %4, %8\nv_bcnt_u32_b32 %5, %5, %8\nv_bcnt_u32_b32 %6, %6, %8\nv_bcnt_u32_b32 %7, %7, %8\n" : ACC8U : "v"(ub)); }) \
    X(CMP_VCC, "v_cmp_lt_f32 vcc, a, b", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\n" \
        "v_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n" : ACC8F : "v"(b) : "vcc"); }) \
    X(CMP_SGPR, "v_cmp_lt_f32 s[2k:2k+1], a, b (8 SGPR pairs)", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_cmp_lt_f32 s[20:21], %0, %8\nv_cmp_lt_f32 s[22:23], %1, %8\nv_cmp_lt_f32 s[24:25], %2, %8\nv_cmp_lt_f32 s[26:27], %3, %8\n" \
        "v_cmp_lt_f32 s[28:29], %4, %8\nv_cmp_lt_f32 s[30:31], %5, %8\nv_cmp_lt_f32 s[32:33], %6, %8\nv_cmp_lt_f32 s[34:35], %7, %8\n" : ACC8F : "v"(b) \
        : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35"); }) \
    X(CMP_ADDC, "pair: v_cmp_lt_f32 vcc + v_addc_co_u32 m, vcc, m, m, vcc  (2 instr)", { _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) asm volatile( \
        "v_cmp_lt_f32 vcc, %0, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\nv_cmp_lt_f32 vcc, %1, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\n" \
        "v_cmp_lt_f32 vcc, %2, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\n" \
        "v_cmp_lt_f32 vcc, %4, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\nv_cmp_lt_f32 vcc, %5, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\n" \
        "v_cmp_lt_f32 vcc, %6, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\nv_cmp_lt_f32 vcc, %7, %8\nv_addc_co_u32 %9, vcc, %9, %9, vcc\n" \
        : ACC8F : "v"(b), "v"(u0) : "vcc"); }) \
    X(CMP_CNDMASK, "pair: v_cmp_lt_f32 vcc + v_cndmask_b32 d, d, b, vcc  (2 instr)", { _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) asm volatile( \
        "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_lt_f32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %8, vcc\n" \
        "v_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %8, vcc\n" \
        "v_cmp_lt_f32 vcc, %4, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_cmp_lt_f32 vcc, %5, %8\nv_cndmask_b32 %5, %5, %8, vcc\n" \
        "v_cmp_lt_f32 vcc, %6, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_cmp_lt_f32 vcc, %7, %8\nv_cndmask_b32 %7, %7, %8, vcc\n" \
        : ACC8F : "v"(b) : "vcc"); }) \
    X(CNDMASK_SGPR, "v_cndmask_b32 d, d, b, s[20:21] (mask set once)", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_cndmask_b32 %0, %0, %8, s[20:21]\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_cndmask_b32 %2, %2, %8, s[20:21]\nv_cndmask_b32 %3, %3, %8, s[20:21]\n" \
        "v_cndmask_b32 %4, %4, %8, s[20:21]\nv_cndmask_b32 %5, %5, %8, s[20:21]\nv_cndmask_b32 %6, %6, %8, s[20:21]\nv_cndmask_b32 %7, %7, %8, s[20:21]\n" \
        : ACC8F : "v"(b) : "s20", "s21"); }) \
    X(FMA_SGPR, "v_fma_f32 d, d, s, c (one SGPR operand)", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n" \
        "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n" : ACC8F : "s"(sb), "v"(c)); }) \
    X(FMAMK, "v_fmamk_f32 d, d, K, c (literal)", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_fmamk_f32 %0, %0, 0x3f7fbe77, %8\nv_fmamk_f32 %1, %1, 0x3f7fbe77, %8\nv_fmamk_f32 %2, %2, 0x3f7fbe77, %8\nv_fmamk_f32 %3, %3, 0x3f7fbe77, %8\n" \
        "v_fmamk_f32 %4, %4, 0x3f7fbe77, %8\nv_fmamk_f32 %5, %5, 0x3f7fbe77, %8\nv_fmamk_f32 %6, %6, 0x3f7fbe77, %8\nv_fmamk_f32 %7, %7, 0x3f7fbe77, %8\n" : ACC8F : "v"(c)); }) \
    X(DOT2_F16, "v_dot2_f32_f16 d, h2, h2, d", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_dot2_f32_f16 %0, %8, %9, %0\nv_dot2_f32_f16 %1, %8, %9, %1\nv_dot2_f32_f16 %2, %8, %9, %2\nv_dot2_f32_f16 %3, %8, %9, %3\n" \
        "v_dot2_f32_f16 %4, %8, %9, %4\nv_dot2_f32_f16 %5, %8, %9, %5\nv_dot2_f32_f16 %6, %8, %9, %6\nv_dot2_f32_f16 %7, %8, %9, %7\n" : ACC8F : "v"(hb), "v"(hc)); }) \
    X(DOT2C_F16, "v_dot2c_f32_f16 d, h2, h2", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_dot2c_f32_f16 %0, %8, %9\nv_dot2c_f32_f16 %1, %8, %9\nv_dot2c_f32_f16 %2, %8, %9\nv_dot2c_f32_f16 %3, %8, %9\n" \
        "v_dot2c_f32_f16 %4, %8, %9\nv_dot2c_f32_f16 %5, %8, %9\nv_dot2c_f32_f16 %6, %8, %9\nv_dot2c_f32_f16 %7, %8, %9\n" : ACC8F : "v"(hb), "v"(hc)); }) \
    X(PK_FMA_F16, "v_pk_fma_f16 d, d, h2, h2", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_pk_fma_f16 %0, %0, %8, %9\nv_pk_fma_f16 %1, %1, %8, %9\nv_pk_fma_f16 %2, %2, %8, %9\nv_pk_fma_f16 %3, %3, %8, %9\n" \
        "v_pk_fma_f16 %4, %4, %8, %9\nv_pk_fma_f16 %5, %5, %8, %9\nv_pk_fma_f16 %6, %6, %8, %9\nv_pk_fma_f16 %7, %7, %8, %9\n" : ACC8U : "v"(hb), "v"(hc)); }) \
    X(LSHL_ADD_U64, "v_lshl_add_u64 d64, d64, 0, s64", { _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) asm volatile( \
        "v_lshl_add_u64 %0, %0, 0, %4\nv_lshl_add_u64 %1, %1, 0, %4\nv_lshl_add_u64 %2, %2, 0, %4\nv_lshl_add_u64 %3, %3, 0, %4\n" \
        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "s"(sw)); }) \
    X(MAD_U64_U32, "v_mad_u64_u32 d64, s, a, b, d64", { _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) asm volatile( \
        "v_mad_u64_u32 %0, s[20:21], %4, %5, %0\nv_mad_u64_u32 %1, s[20:21], %4, %5, %1\nv_mad_u64_u32 %2, s[20:21], %4, %5, %2\nv_mad_u64_u32 %3, s[20:21], %4, %5, %3\n" \
        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(ub), "v"(uc) : "s20", "s21"); }) \
    X(READLANE, "v_readlane_b32 s, v, 5 (8 SGPRs)", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_readlane_b32 s20, %0, 5\nv_readlane_b32 s21, %1, 5\nv_readlane_b32 s22, %2, 5\nv_readlane_b32 s23, %3, 5\n" \
        "v_readlane_b32 s24, %4, 5\nv_readlane_b32 s25, %5, 5\nv_readlane_b32 s26, %6, 5\nv_readlane_b32 s27, %7, 5\n" : ACC8F : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); }) \
    X(MOV_DPP, "v_mov_b32_dpp d, d row_shr:1", { _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) asm volatile( \
        "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" \
        "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n" : ACC8F); }) \
    X(DS_READ_B64, "ds_read_b64 (conflict-free, 16 in flight)", { for (int r_ = 0; r_ < 4; ++r_) { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) \
        asm volatile("ds_read_b64 %0, %2\nds_read_b64 %1, %2 offset:1024\n" : "=v"(w0), "=v"(w1) : "v"(addr8)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } }) \
    X(DS_READ_B96, "ds_read_b96 (16-B records, 16 in flight)", { for (int r_ = 0; r_ < 4; ++r_) { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) \
        asm volatile("ds_read_b96 %0, %2\nds_read_b96 %1, %2 offset:1024\n" : "=v"(t0), "=v"(t1) : "v"(addr16)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } }) \
    X(DS_READ_B128_BCAST8, "ds_read_b128, 8 lanes share an address (the filter's pattern)", { for (int r_ = 0; r_ < 4; ++r_) { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) \
        asm volatile("ds_read_b128 %0, %2\nds_read_b128 %1, %2 offset:1024\n" : "=v"(q0), "=v"(q1) : "v"(addrb)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } }) \
    X(DS_READ_U16, "ds_read_u16 (16 in flight)", { for (int r_ = 0; r_ < 4; ++r_) { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) \
        asm volatile("ds_read_u16 %0, %2\nds_read_u16 %1, %2 offset:1024\n" : "=v"(u0), "=v"(u1) : "v"(addr2)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } }) \
    X(DS_WRITE_B16, "ds_write_b16 (16 in flight)", { for (int r_ = 0; r_ < 4; ++r_) { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) \
        asm volatile("ds_write_b16 %0, %1\nds_write_b16 %0, %1 offset:1024\n" : : "v"(addr2), "v"(u0)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } }) \
    X(DS_WRITE_B32, "ds_write_b32 (16 in flight)", { for (int r_ = 0; r_ < 4; ++r_) { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) \
        asm volatile("ds_write_b32 %0, %1\nds_write_b32 %0, %1 offset:1024\n" : : "v"(addr4), "v"(u0)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } }) \
    X(DS_BPERMUTE, "ds_bpermute_b32 (16 in flight)", { for (int r_ = 0; r_ < 4; ++r_) { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) \
        asm volatile("ds_bpermute_b32 %0, %2, %3\nds_bpermute_b32 %1, %2, %3\n" : "=v"(u0), "=v"(u1) : "v"(addr4), "v"(ub)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } })

enum {
#define X(ID, LABEL, BODY) OP_##ID,
    OPS
#undef X
    OP_COUNT
};
static const char* op_label[] = {
#define X(ID, LABEL, BODY) LABEL,
    OPS
#undef X
};

template <int OP>
__global__ __launch_bounds__(256) void k_bench(unsigned long long* cyc, float* sink, int iters, float sb, unsigned long long sw) {
    __shared__ float4v lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) lds[i] = float4v{(float)i, 1.0f, 2.0f, 3.0f};
    __syncthreads();
    float a0 = tid * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.999f, c = 1e-3f;
    unsigned u0 = tid * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned ub = 3u, uc = 5u;
    half2v hb = {(_Float16)0.5f, (_Float16)0.25f}, hc = {(_Float16)1.5f, (_Float16)0.75f};
    unsigned long long w0 = tid, w1 = tid + 1, w2 = tid + 2, w3 = tid + 3;
    float4v q0 = {}, q1 = {};
    typedef float float3v __attribute__((ext_vector_type(3)));
    float3v t0 = {}, t1 = {};
    const unsigned lane = tid & 63;
    unsigned addr16 = lane * 16u, addr8 = lane * 8u, addr4 = lane * 4u, addr2 = lane * 2u;
    unsigned addrb = (lane >> 3) * 128u;  // 8 cells' runs start 8 records apart; the 8 lanes of a cell read one address
    asm volatile("s_mov_b32 s20, 0x55555555\ns_mov_b32 s21, 0x55555555" ::: "s20", "s21");
    const unsigned long long t_begin = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define X(ID, LABEL, BODY) if (OP == OP_##ID) BODY
        OPS
#undef X
    }
    const unsigned long long t_end = __builtin_readcyclecounter();
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t_end - t_begin;
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(w0 ^ w1 ^ w2 ^ w3) + q0.x + q1.y + t0.x + t1.y;
    if (s == 123.456f) sink[tid] = s;
}

template <int OP>
static void run_op(unsigned long long* d_cyc, float* d_sink) {
    const int iters = 1000;
    for (int k : {2, 4}) {
        const int grid = 256 * k;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_bench<OP>, dim3(grid), dim3(256), 0, 0, d_cyc, d_sink, 20, 0.999f, 128ull);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bench<OP>, dim3(grid), dim3(256), 0, 0, d_cyc, d_sink, iters, 0.999f, 128ull);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid * 4);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2], mn = (double)h.front();
        const double n_inst = (double)iters * 64;
        const double ns_per_inst_simd = ms * 1e6 / (n_inst * grid * 4 / 1024.0);
        printf("%-72s k=%d  cyc/inst/wave med %6.2f min %6.2f  cyc/inst/SIMD %5.2f   event: %5.2f ns per wave-inst per SIMD\n", op_label[OP], k,
               med / n_inst, mn / n_inst, med / n_inst / k, ns_per_inst_simd);
        fflush(stdout);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

template <int OP>
static void run_all(unsigned long long* d_cyc, float* d_sink) {
    run_op<OP>(d_cyc, d_sink);
    if constexpr (OP + 1 < OP_COUNT) run_all<OP + 1>(d_cyc, d_sink);
}

int main() {
    unsigned long long* d_cyc;
    float* d_sink;
    CHECK(hipMalloc(&d_cyc, 256 * 8 * 4 * 8));
    CHECK(hipMalloc(&d_sink, 256 * 4));
    printf("# 64 instructions x 1000 trips per wave; 256*k workgroups of 4 waves; event column = wall time per wave-instruction per SIMD\n");
    printf("# (at ~2.2 GHz under load 1 ns ~ 2.2 cycles; LDS rows: all 4 SIMDs of a CU share one LDS pipe)\n");
    run_all<0>(d_cyc, d_sink);
    return 0;
}
