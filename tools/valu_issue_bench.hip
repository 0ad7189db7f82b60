// valu_issue_bench.hip — issue rate of independent FP32 VALU instructions on gfx950, by
// instruction form and operand source (inline asm, 32 independent destinations per loop
// trip, so neither dependencies nor the compiler shape the result).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_bench.hip -o tools/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// Each body is 16 independent instructions writing v[40..71]; sources v[8..15], s[20..23].
#define BODY_MUL_VV \
    "v_mul_f32 v40, v8, v9\n v_mul_f32 v41, v9, v10\n v_mul_f32 v42, v10, v11\n v_mul_f32 v43, v11, v12\n" \
    "v_mul_f32 v44, v12, v13\n v_mul_f32 v45, v13, v14\n v_mul_f32 v46, v14, v15\n v_mul_f32 v47, v15, v8\n" \
    "v_mul_f32 v48, v8, v10\n v_mul_f32 v49, v9, v11\n v_mul_f32 v50, v10, v12\n v_mul_f32 v51, v11, v13\n" \
    "v_mul_f32 v52, v12, v14\n v_mul_f32 v53, v13, v15\n v_mul_f32 v54, v14, v8\n v_mul_f32 v55, v15, v9\n"
#define BODY_MUL_SV \
    "v_mul_f32 v40, s20, v9\n v_mul_f32 v41, s21, v10\n v_mul_f32 v42, s22, v11\n v_mul_f32 v43, s23, v12\n" \
    "v_mul_f32 v44, s20, v13\n v_mul_f32 v45, s21, v14\n v_mul_f32 v46, s22, v15\n v_mul_f32 v47, s23, v8\n" \
    "v_mul_f32 v48, s20, v10\n v_mul_f32 v49, s21, v11\n v_mul_f32 v50, s22, v12\n v_mul_f32 v51, s23, v13\n" \
    "v_mul_f32 v52, s20, v14\n v_mul_f32 v53, s21, v15\n v_mul_f32 v54, s22, v8\n v_mul_f32 v55, s23, v9\n"
#define BODY_ADD_VV \
    "v_add_f32 v40, v8, v9\n v_add_f32 v41, v9, v10\n v_add_f32 v42, v10, v11\n v_add_f32 v43, v11, v12\n" \
    "v_add_f32 v44, v12, v13\n v_add_f32 v45, v13, v14\n v_add_f32 v46, v14, v15\n v_add_f32 v47, v15, v8\n" \
    "v_add_f32 v48, v8, v10\n v_add_f32 v49, v9, v11\n v_add_f32 v50, v10, v12\n v_add_f32 v51, v11, v13\n" \
    "v_add_f32 v52, v12, v14\n v_add_f32 v53, v13, v15\n v_add_f32 v54, v14, v8\n v_add_f32 v55, v15, v9\n"
#define BODY_FMA_VVV \
    "v_fma_f32 v40, v8, v9, v10\n v_fma_f32 v41, v9, v10, v11\n v_fma_f32 v42, v10, v11, v12\n v_fma_f32 v43, v11, v12, v13\n" \
    "v_fma_f32 v44, v12, v13, v14\n v_fma_f32 v45, v13, v14, v15\n v_fma_f32 v46, v14, v15, v8\n v_fma_f32 v47, v15, v8, v9\n" \
    "v_fma_f32 v48, v8, v10, v12\n v_fma_f32 v49, v9, v11, v13\n v_fma_f32 v50, v10, v12, v14\n v_fma_f32 v51, v11, v13, v15\n" \
    "v_fma_f32 v52, v12, v14, v8\n v_fma_f32 v53, v13, v15, v9\n v_fma_f32 v54, v14, v8, v10\n v_fma_f32 v55, v15, v9, v11\n"
#define BODY_PKMUL_VV \
    "v_pk_mul_f32 v[40:41], v[8:9], v[10:11]\n v_pk_mul_f32 v[42:43], v[10:11], v[12:13]\n v_pk_mul_f32 v[44:45], v[12:13], v[14:15]\n v_pk_mul_f32 v[46:47], v[14:15], v[8:9]\n" \
    "v_pk_mul_f32 v[48:49], v[8:9], v[12:13]\n v_pk_mul_f32 v[50:51], v[10:11], v[14:15]\n v_pk_mul_f32 v[52:53], v[12:13], v[8:9]\n v_pk_mul_f32 v[54:55], v[14:15], v[10:11]\n" \
    "v_pk_mul_f32 v[56:57], v[8:9], v[14:15]\n v_pk_mul_f32 v[58:59], v[10:11], v[8:9]\n v_pk_mul_f32 v[60:61], v[12:13], v[10:11]\n v_pk_mul_f32 v[62:63], v[14:15], v[12:13]\n" \
    "v_pk_mul_f32 v[64:65], v[8:9], v[8:9]\n v_pk_mul_f32 v[66:67], v[10:11], v[10:11]\n v_pk_mul_f32 v[68:69], v[12:13], v[12:13]\n v_pk_mul_f32 v[70:71], v[14:15], v[14:15]\n"
#define BODY_PKMUL_VS \
    "v_pk_mul_f32 v[40:41], v[8:9], s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 v[42:43], v[10:11], s[22:23] op_sel_hi:[1,0]\n v_pk_mul_f32 v[44:45], v[12:13], s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 v[46:47], v[14:15], s[22:23] op_sel_hi:[1,0]\n" \
    "v_pk_mul_f32 v[48:49], v[8:9], s[22:23] op_sel_hi:[1,0]\n v_pk_mul_f32 v[50:51], v[10:11], s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 v[52:53], v[12:13], s[22:23] op_sel_hi:[1,0]\n v_pk_mul_f32 v[54:55], v[14:15], s[20:21] op_sel_hi:[1,0]\n" \
    "v_pk_mul_f32 v[56:57], v[8:9], s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 v[58:59], v[10:11], s[22:23] op_sel_hi:[1,0]\n v_pk_mul_f32 v[60:61], v[12:13], s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 v[62:63], v[14:15], s[22:23] op_sel_hi:[1,0]\n" \
    "v_pk_mul_f32 v[64:65], v[8:9], s[22:23] op_sel_hi:[1,0]\n v_pk_mul_f32 v[66:67], v[10:11], s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 v[68:69], v[12:13], s[22:23] op_sel_hi:[1,0]\n v_pk_mul_f32 v[70:71], v[14:15], s[20:21] op_sel_hi:[1,0]\n"
#define BODY_PKFMA_VVV \
    "v_pk_fma_f32 v[40:41], v[8:9], v[10:11], v[12:13]\n v_pk_fma_f32 v[42:43], v[10:11], v[12:13], v[14:15]\n v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[8:9]\n v_pk_fma_f32 v[46:47], v[14:15], v[8:9], v[10:11]\n" \
    "v_pk_fma_f32 v[48:49], v[8:9], v[12:13], v[10:11]\n v_pk_fma_f32 v[50:51], v[10:11], v[14:15], v[12:13]\n v_pk_fma_f32 v[52:53], v[12:13], v[8:9], v[14:15]\n v_pk_fma_f32 v[54:55], v[14:15], v[10:11], v[8:9]\n" \
    "v_pk_fma_f32 v[56:57], v[8:9], v[14:15], v[10:11]\n v_pk_fma_f32 v[58:59], v[10:11], v[8:9], v[12:13]\n v_pk_fma_f32 v[60:61], v[12:13], v[10:11], v[14:15]\n v_pk_fma_f32 v[62:63], v[14:15], v[12:13], v[8:9]\n" \
    "v_pk_fma_f32 v[64:65], v[8:9], v[8:9], v[10:11]\n v_pk_fma_f32 v[66:67], v[10:11], v[10:11], v[12:13]\n v_pk_fma_f32 v[68:69], v[12:13], v[12:13], v[14:15]\n v_pk_fma_f32 v[70:71], v[14:15], v[14:15], v[8:9]\n"
#define BODY_CMP \
    "v_cmp_ge_f32 vcc, v8, v9\n v_cmp_ge_f32 vcc, v9, v10\n v_cmp_ge_f32 vcc, v10, v11\n v_cmp_ge_f32 vcc, v11, v12\n" \
    "v_cmp_ge_f32 vcc, v12, v13\n v_cmp_ge_f32 vcc, v13, v14\n v_cmp_ge_f32 vcc, v14, v15\n v_cmp_ge_f32 vcc, v15, v8\n" \
    "v_cmp_ge_f32 vcc, v8, v10\n v_cmp_ge_f32 vcc, v9, v11\n v_cmp_ge_f32 vcc, v10, v12\n v_cmp_ge_f32 vcc, v11, v13\n" \
    "v_cmp_ge_f32 vcc, v12, v14\n v_cmp_ge_f32 vcc, v13, v15\n v_cmp_ge_f32 vcc, v14, v8\n v_cmp_ge_f32 vcc, v15, v9\n"
#define BODY_MAX3 \
    "v_max3_f32 v40, v8, v9, v10\n v_max3_f32 v41, v9, v10, v11\n v_max3_f32 v42, v10, v11, v12\n v_max3_f32 v43, v11, v12, v13\n" \
    "v_max3_f32 v44, v12, v13, v14\n v_max3_f32 v45, v13, v14, v15\n v_max3_f32 v46, v14, v15, v8\n v_max3_f32 v47, v15, v8, v9\n" \
    "v_max3_f32 v48, v8, v10, v12\n v_max3_f32 v49, v9, v11, v13\n v_max3_f32 v50, v10, v12, v14\n v_max3_f32 v51, v11, v13, v15\n" \
    "v_max3_f32 v52, v12, v14, v8\n v_max3_f32 v53, v13, v15, v9\n v_max3_f32 v54, v14, v8, v10\n v_max3_f32 v55, v15, v9, v11\n"

#define BODY_MIN_VV \
    "v_min_f32 v40, v8, v9\n v_min_f32 v41, v9, v10\n v_min_f32 v42, v10, v11\n v_min_f32 v43, v11, v12\n" \
    "v_min_f32 v44, v12, v13\n v_min_f32 v45, v13, v14\n v_min_f32 v46, v14, v15\n v_min_f32 v47, v15, v8\n" \
    "v_min_f32 v48, v8, v9\n v_min_f32 v49, v9, v10\n v_min_f32 v50, v10, v11\n v_min_f32 v51, v11, v12\n" \
    "v_min_f32 v52, v12, v13\n v_min_f32 v53, v13, v14\n v_min_f32 v54, v14, v15\n v_min_f32 v55, v15, v8\n"
#define BODY_MAX_VV \
    "v_max_f32 v40, v8, v9\n v_max_f32 v41, v9, v10\n v_max_f32 v42, v10, v11\n v_max_f32 v43, v11, v12\n" \
    "v_max_f32 v44, v12, v13\n v_max_f32 v45, v13, v14\n v_max_f32 v46, v14, v15\n v_max_f32 v47, v15, v8\n" \
    "v_max_f32 v48, v8, v9\n v_max_f32 v49, v9, v10\n v_max_f32 v50, v10, v11\n v_max_f32 v51, v11, v12\n" \
    "v_max_f32 v52, v12, v13\n v_max_f32 v53, v13, v14\n v_max_f32 v54, v14, v15\n v_max_f32 v55, v15, v8\n"
#define BODY_SUB_VV \
    "v_sub_f32 v40, v8, v9\n v_sub_f32 v41, v9, v10\n v_sub_f32 v42, v10, v11\n v_sub_f32 v43, v11, v12\n" \
    "v_sub_f32 v44, v12, v13\n v_sub_f32 v45, v13, v14\n v_sub_f32 v46, v14, v15\n v_sub_f32 v47, v15, v8\n" \
    "v_sub_f32 v48, v8, v9\n v_sub_f32 v49, v9, v10\n v_sub_f32 v50, v10, v11\n v_sub_f32 v51, v11, v12\n" \
    "v_sub_f32 v52, v12, v13\n v_sub_f32 v53, v13, v14\n v_sub_f32 v54, v14, v15\n v_sub_f32 v55, v15, v8\n"
#define BODY_MIN_NEG \
    "v_min_f32_e64 v40, v8, -v9\n v_min_f32_e64 v41, v9, -v10\n v_min_f32_e64 v42, v10, -v11\n v_min_f32_e64 v43, v11, -v12\n" \
    "v_min_f32_e64 v44, v12, -v13\n v_min_f32_e64 v45, v13, -v14\n v_min_f32_e64 v46, v14, -v15\n v_min_f32_e64 v47, v15, -v8\n" \
    "v_min_f32_e64 v48, v8, -v9\n v_min_f32_e64 v49, v9, -v10\n v_min_f32_e64 v50, v10, -v11\n v_min_f32_e64 v51, v11, -v12\n" \
    "v_min_f32_e64 v52, v12, -v13\n v_min_f32_e64 v53, v13, -v14\n v_min_f32_e64 v54, v14, -v15\n v_min_f32_e64 v55, v15, -v8\n"
#define BODY_MUL_ABS \
    "v_mul_f32_e64 v40, v8, -|v9|\n v_mul_f32_e64 v41, v9, -|v10|\n v_mul_f32_e64 v42, v10, -|v11|\n v_mul_f32_e64 v43, v11, -|v12|\n" \
    "v_mul_f32_e64 v44, v12, -|v13|\n v_mul_f32_e64 v45, v13, -|v14|\n v_mul_f32_e64 v46, v14, -|v15|\n v_mul_f32_e64 v47, v15, -|v8|\n" \
    "v_mul_f32_e64 v48, v8, -|v9|\n v_mul_f32_e64 v49, v9, -|v10|\n v_mul_f32_e64 v50, v10, -|v11|\n v_mul_f32_e64 v51, v11, -|v12|\n" \
    "v_mul_f32_e64 v52, v12, -|v13|\n v_mul_f32_e64 v53, v13, -|v14|\n v_mul_f32_e64 v54, v14, -|v15|\n v_mul_f32_e64 v55, v15, -|v8|\n"
#define BODY_FMA_NEG \
    "v_fma_f32 v40, v8, v9, -v11\n v_fma_f32 v41, v9, v10, -v12\n v_fma_f32 v42, v10, v11, -v13\n v_fma_f32 v43, v11, v12, -v14\n" \
    "v_fma_f32 v44, v12, v13, -v15\n v_fma_f32 v45, v13, v14, -v8\n v_fma_f32 v46, v14, v15, -v9\n v_fma_f32 v47, v15, v8, -v10\n" \
    "v_fma_f32 v48, v8, v9, -v11\n v_fma_f32 v49, v9, v10, -v12\n v_fma_f32 v50, v10, v11, -v13\n v_fma_f32 v51, v11, v12, -v14\n" \
    "v_fma_f32 v52, v12, v13, -v15\n v_fma_f32 v53, v13, v14, -v8\n v_fma_f32 v54, v14, v15, -v9\n v_fma_f32 v55, v15, v8, -v10\n"
#define BODY_FMA_S \
    "v_fma_f32 v40, v8, s20, v11\n v_fma_f32 v41, v9, s20, v12\n v_fma_f32 v42, v10, s20, v13\n v_fma_f32 v43, v11, s20, v14\n" \
    "v_fma_f32 v44, v12, s20, v15\n v_fma_f32 v45, v13, s20, v8\n v_fma_f32 v46, v14, s20, v9\n v_fma_f32 v47, v15, s20, v10\n" \
    "v_fma_f32 v48, v8, s20, v11\n v_fma_f32 v49, v9, s20, v12\n v_fma_f32 v50, v10, s20, v13\n v_fma_f32 v51, v11, s20, v14\n" \
    "v_fma_f32 v52, v12, s20, v15\n v_fma_f32 v53, v13, s20, v8\n v_fma_f32 v54, v14, s20, v9\n v_fma_f32 v55, v15, s20, v10\n"
#define BODY_FMA_K \
    "v_fma_f32 v40, v8, v9, 1.0\n v_fma_f32 v41, v9, v10, 1.0\n v_fma_f32 v42, v10, v11, 1.0\n v_fma_f32 v43, v11, v12, 1.0\n" \
    "v_fma_f32 v44, v12, v13, 1.0\n v_fma_f32 v45, v13, v14, 1.0\n v_fma_f32 v46, v14, v15, 1.0\n v_fma_f32 v47, v15, v8, 1.0\n" \
    "v_fma_f32 v48, v8, v9, 1.0\n v_fma_f32 v49, v9, v10, 1.0\n v_fma_f32 v50, v10, v11, 1.0\n v_fma_f32 v51, v11, v12, 1.0\n" \
    "v_fma_f32 v52, v12, v13, 1.0\n v_fma_f32 v53, v13, v14, 1.0\n v_fma_f32 v54, v14, v15, 1.0\n v_fma_f32 v55, v15, v8, 1.0\n"
#define BODY_AND \
    "v_and_b32 v40, v8, v9\n v_and_b32 v41, v9, v10\n v_and_b32 v42, v10, v11\n v_and_b32 v43, v11, v12\n" \
    "v_and_b32 v44, v12, v13\n v_and_b32 v45, v13, v14\n v_and_b32 v46, v14, v15\n v_and_b32 v47, v15, v8\n" \
    "v_and_b32 v48, v8, v9\n v_and_b32 v49, v9, v10\n v_and_b32 v50, v10, v11\n v_and_b32 v51, v11, v12\n" \
    "v_and_b32 v52, v12, v13\n v_and_b32 v53, v13, v14\n v_and_b32 v54, v14, v15\n v_and_b32 v55, v15, v8\n"
#define BODY_OR \
    "v_or_b32 v40, v8, v9\n v_or_b32 v41, v9, v10\n v_or_b32 v42, v10, v11\n v_or_b32 v43, v11, v12\n" \
    "v_or_b32 v44, v12, v13\n v_or_b32 v45, v13, v14\n v_or_b32 v46, v14, v15\n v_or_b32 v47, v15, v8\n" \
    "v_or_b32 v48, v8, v9\n v_or_b32 v49, v9, v10\n v_or_b32 v50, v10, v11\n v_or_b32 v51, v11, v12\n" \
    "v_or_b32 v52, v12, v13\n v_or_b32 v53, v13, v14\n v_or_b32 v54, v14, v15\n v_or_b32 v55, v15, v8\n"
#define BODY_CNDMASK \
    "v_cndmask_b32 v40, v8, v9, vcc\n v_cndmask_b32 v41, v9, v10, vcc\n v_cndmask_b32 v42, v10, v11, vcc\n v_cndmask_b32 v43, v11, v12, vcc\n" \
    "v_cndmask_b32 v44, v12, v13, vcc\n v_cndmask_b32 v45, v13, v14, vcc\n v_cndmask_b32 v46, v14, v15, vcc\n v_cndmask_b32 v47, v15, v8, vcc\n" \
    "v_cndmask_b32 v48, v8, v9, vcc\n v_cndmask_b32 v49, v9, v10, vcc\n v_cndmask_b32 v50, v10, v11, vcc\n v_cndmask_b32 v51, v11, v12, vcc\n" \
    "v_cndmask_b32 v52, v12, v13, vcc\n v_cndmask_b32 v53, v13, v14, vcc\n v_cndmask_b32 v54, v14, v15, vcc\n v_cndmask_b32 v55, v15, v8, vcc\n"
#define BODY_MOV \
    "v_mov_b32 v40, v8\n v_mov_b32 v41, v9\n v_mov_b32 v42, v10\n v_mov_b32 v43, v11\n" \
    "v_mov_b32 v44, v12\n v_mov_b32 v45, v13\n v_mov_b32 v46, v14\n v_mov_b32 v47, v15\n" \
    "v_mov_b32 v48, v8\n v_mov_b32 v49, v9\n v_mov_b32 v50, v10\n v_mov_b32 v51, v11\n" \
    "v_mov_b32 v52, v12\n v_mov_b32 v53, v13\n v_mov_b32 v54, v14\n v_mov_b32 v55, v15\n"
#define BODY_SUBREV_S \
    "v_subrev_f32 v40, s21, v9\n v_subrev_f32 v41, s21, v10\n v_subrev_f32 v42, s21, v11\n v_subrev_f32 v43, s21, v12\n" \
    "v_subrev_f32 v44, s21, v13\n v_subrev_f32 v45, s21, v14\n v_subrev_f32 v46, s21, v15\n v_subrev_f32 v47, s21, v8\n" \
    "v_subrev_f32 v48, s21, v9\n v_subrev_f32 v49, s21, v10\n v_subrev_f32 v50, s21, v11\n v_subrev_f32 v51, s21, v12\n" \
    "v_subrev_f32 v52, s21, v13\n v_subrev_f32 v53, s21, v14\n v_subrev_f32 v54, s21, v15\n v_subrev_f32 v55, s21, v8\n"
#define BODY_MAC \
    "v_fmac_f32 v40, v8, v9\n v_fmac_f32 v41, v9, v10\n v_fmac_f32 v42, v10, v11\n v_fmac_f32 v43, v11, v12\n" \
    "v_fmac_f32 v44, v12, v13\n v_fmac_f32 v45, v13, v14\n v_fmac_f32 v46, v14, v15\n v_fmac_f32 v47, v15, v8\n" \
    "v_fmac_f32 v48, v8, v9\n v_fmac_f32 v49, v9, v10\n v_fmac_f32 v50, v10, v11\n v_fmac_f32 v51, v11, v12\n" \
    "v_fmac_f32 v52, v12, v13\n v_fmac_f32 v53, v13, v14\n v_fmac_f32 v54, v14, v15\n v_fmac_f32 v55, v15, v8\n"
#define BODY_FMA_2BANK \
    "v_fma_f32 v40, v8, v12, v9\n v_fma_f32 v41, v12, v8, v10\n v_fma_f32 v42, v8, v12, v11\n v_fma_f32 v43, v12, v8, v9\n" \
    "v_fma_f32 v44, v8, v12, v10\n v_fma_f32 v45, v12, v8, v11\n v_fma_f32 v46, v8, v12, v9\n v_fma_f32 v47, v12, v8, v10\n" \
    "v_fma_f32 v48, v8, v12, v11\n v_fma_f32 v49, v12, v8, v9\n v_fma_f32 v50, v8, v12, v10\n v_fma_f32 v51, v12, v8, v11\n" \
    "v_fma_f32 v52, v8, v12, v9\n v_fma_f32 v53, v12, v8, v10\n v_fma_f32 v54, v8, v12, v11\n v_fma_f32 v55, v12, v8, v9\n"
#define BODY_FMA_2BANKB \
    "v_fma_f32 v40, v9, v8, v12\n v_fma_f32 v41, v10, v8, v12\n v_fma_f32 v42, v11, v8, v12\n v_fma_f32 v43, v9, v8, v12\n" \
    "v_fma_f32 v44, v10, v8, v12\n v_fma_f32 v45, v11, v8, v12\n v_fma_f32 v46, v9, v8, v12\n v_fma_f32 v47, v10, v8, v12\n" \
    "v_fma_f32 v48, v11, v8, v12\n v_fma_f32 v49, v9, v8, v12\n v_fma_f32 v50, v10, v8, v12\n v_fma_f32 v51, v11, v8, v12\n" \
    "v_fma_f32 v52, v9, v8, v12\n v_fma_f32 v53, v10, v8, v12\n v_fma_f32 v54, v11, v8, v12\n v_fma_f32 v55, v9, v8, v12\n"
#define BODY_FMA_NOCONF \
    "v_fma_f32 v40, v8, v9, v10\n v_fma_f32 v41, v8, v9, v11\n v_fma_f32 v42, v8, v9, v10\n v_fma_f32 v43, v8, v9, v11\n" \
    "v_fma_f32 v44, v8, v9, v10\n v_fma_f32 v45, v8, v9, v11\n v_fma_f32 v46, v8, v9, v10\n v_fma_f32 v47, v8, v9, v11\n" \
    "v_fma_f32 v48, v8, v9, v10\n v_fma_f32 v49, v8, v9, v11\n v_fma_f32 v50, v8, v9, v10\n v_fma_f32 v51, v8, v9, v11\n" \
    "v_fma_f32 v52, v8, v9, v10\n v_fma_f32 v53, v8, v9, v11\n v_fma_f32 v54, v8, v9, v10\n v_fma_f32 v55, v8, v9, v11\n"
#define BODY_MUL_2BANK \
    "v_mul_f32 v40, v8, v12\n v_mul_f32 v41, v8, v12\n v_mul_f32 v42, v8, v12\n v_mul_f32 v43, v8, v12\n" \
    "v_mul_f32 v44, v8, v12\n v_mul_f32 v45, v8, v12\n v_mul_f32 v46, v8, v12\n v_mul_f32 v47, v8, v12\n" \
    "v_mul_f32 v48, v8, v12\n v_mul_f32 v49, v8, v12\n v_mul_f32 v50, v8, v12\n v_mul_f32 v51, v8, v12\n" \
    "v_mul_f32 v52, v8, v12\n v_mul_f32 v53, v8, v12\n v_mul_f32 v54, v8, v12\n v_mul_f32 v55, v8, v12\n"
#define BODY_FMAC_S01 \
    "v_fmac_f32 v40, v8, v12\n v_fmac_f32 v41, v8, v12\n v_fmac_f32 v42, v8, v12\n v_fmac_f32 v43, v8, v12\n" \
    "v_fmac_f32 v44, v8, v12\n v_fmac_f32 v45, v8, v12\n v_fmac_f32 v46, v8, v12\n v_fmac_f32 v47, v8, v12\n" \
    "v_fmac_f32 v48, v8, v12\n v_fmac_f32 v49, v8, v12\n v_fmac_f32 v50, v8, v12\n v_fmac_f32 v51, v8, v12\n" \
    "v_fmac_f32 v52, v8, v12\n v_fmac_f32 v53, v8, v12\n v_fmac_f32 v54, v8, v12\n v_fmac_f32 v55, v8, v12\n"
#define BODY_FMAC_D0 \
    "v_fmac_f32 v40, v8, v9\n v_fmac_f32 v44, v8, v9\n v_fmac_f32 v48, v8, v9\n v_fmac_f32 v52, v8, v9\n" \
    "v_fmac_f32 v40, v8, v9\n v_fmac_f32 v44, v8, v9\n v_fmac_f32 v48, v8, v9\n v_fmac_f32 v52, v8, v9\n" \
    "v_fmac_f32 v40, v8, v9\n v_fmac_f32 v44, v8, v9\n v_fmac_f32 v48, v8, v9\n v_fmac_f32 v52, v8, v9\n" \
    "v_fmac_f32 v40, v8, v9\n v_fmac_f32 v44, v8, v9\n v_fmac_f32 v48, v8, v9\n v_fmac_f32 v52, v8, v9\n"
#define BODY_FMAC_D0i \
    "v_fmac_f32 v40, v8, v9\n v_fmac_f32 v44, v8, v9\n v_fmac_f32 v48, v8, v9\n v_fmac_f32 v52, v8, v9\n" \
    "v_fmac_f32 v56, v8, v9\n v_fmac_f32 v60, v8, v9\n v_fmac_f32 v64, v8, v9\n v_fmac_f32 v68, v8, v9\n" \
    "v_fmac_f32 v40, v8, v9\n v_fmac_f32 v44, v8, v9\n v_fmac_f32 v48, v8, v9\n v_fmac_f32 v52, v8, v9\n" \
    "v_fmac_f32 v56, v8, v9\n v_fmac_f32 v60, v8, v9\n v_fmac_f32 v64, v8, v9\n v_fmac_f32 v68, v8, v9\n"
#define BODY_FMAC_D1i \
    "v_fmac_f32 v40, v9, v8\n v_fmac_f32 v44, v9, v8\n v_fmac_f32 v48, v9, v8\n v_fmac_f32 v52, v9, v8\n" \
    "v_fmac_f32 v56, v9, v8\n v_fmac_f32 v60, v9, v8\n v_fmac_f32 v64, v9, v8\n v_fmac_f32 v68, v9, v8\n" \
    "v_fmac_f32 v40, v9, v8\n v_fmac_f32 v44, v9, v8\n v_fmac_f32 v48, v9, v8\n v_fmac_f32 v52, v9, v8\n" \
    "v_fmac_f32 v56, v9, v8\n v_fmac_f32 v60, v9, v8\n v_fmac_f32 v64, v9, v8\n v_fmac_f32 v68, v9, v8\n"
#define BODY_FMAC_OK \
    "v_fmac_f32 v40, v9, v10\n v_fmac_f32 v44, v9, v10\n v_fmac_f32 v48, v9, v10\n v_fmac_f32 v52, v9, v10\n" \
    "v_fmac_f32 v56, v9, v10\n v_fmac_f32 v60, v9, v10\n v_fmac_f32 v64, v9, v10\n v_fmac_f32 v68, v9, v10\n" \
    "v_fmac_f32 v40, v9, v10\n v_fmac_f32 v44, v9, v10\n v_fmac_f32 v48, v9, v10\n v_fmac_f32 v52, v9, v10\n" \
    "v_fmac_f32 v56, v9, v10\n v_fmac_f32 v60, v9, v10\n v_fmac_f32 v64, v9, v10\n v_fmac_f32 v68, v9, v10\n"
#define BODY_FMA_SQ \
    "v_fma_f32 v40, v9, v9, v10\n v_fma_f32 v41, v9, v9, v10\n v_fma_f32 v42, v9, v9, v10\n v_fma_f32 v43, v9, v9, v10\n" \
    "v_fma_f32 v44, v9, v9, v10\n v_fma_f32 v45, v9, v9, v10\n v_fma_f32 v46, v9, v9, v10\n v_fma_f32 v47, v9, v9, v10\n" \
    "v_fma_f32 v48, v9, v9, v10\n v_fma_f32 v49, v9, v9, v10\n v_fma_f32 v50, v9, v9, v10\n v_fma_f32 v51, v9, v9, v10\n" \
    "v_fma_f32 v52, v9, v9, v10\n v_fma_f32 v53, v9, v9, v10\n v_fma_f32 v54, v9, v9, v10\n v_fma_f32 v55, v9, v9, v10\n"
#define BODY_FMA_SQ_CONF \
    "v_fma_f32 v40, v9, v9, v13\n v_fma_f32 v41, v9, v9, v13\n v_fma_f32 v42, v9, v9, v13\n v_fma_f32 v43, v9, v9, v13\n" \
    "v_fma_f32 v44, v9, v9, v13\n v_fma_f32 v45, v9, v9, v13\n v_fma_f32 v46, v9, v9, v13\n v_fma_f32 v47, v9, v9, v13\n" \
    "v_fma_f32 v48, v9, v9, v13\n v_fma_f32 v49, v9, v9, v13\n v_fma_f32 v50, v9, v9, v13\n v_fma_f32 v51, v9, v9, v13\n" \
    "v_fma_f32 v52, v9, v9, v13\n v_fma_f32 v53, v9, v9, v13\n v_fma_f32 v54, v9, v9, v13\n v_fma_f32 v55, v9, v9, v13\n"
#define BODY_BITOP3 \
    "v_bitop3_b32 v40, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v41, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v42, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v43, v8, v9, v10 bitop3:0xc8\n" \
    "v_bitop3_b32 v44, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v45, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v46, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v47, v8, v9, v10 bitop3:0xc8\n" \
    "v_bitop3_b32 v48, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v49, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v50, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v51, v8, v9, v10 bitop3:0xc8\n" \
    "v_bitop3_b32 v52, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v53, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v54, v8, v9, v10 bitop3:0xc8\n v_bitop3_b32 v55, v8, v9, v10 bitop3:0xc8\n"
#define BODY_BITOP3_CONF \
    "v_bitop3_b32 v40, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v41, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v42, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v43, v8, v12, v10 bitop3:0xc8\n" \
    "v_bitop3_b32 v44, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v45, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v46, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v47, v8, v12, v10 bitop3:0xc8\n" \
    "v_bitop3_b32 v48, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v49, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v50, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v51, v8, v12, v10 bitop3:0xc8\n" \
    "v_bitop3_b32 v52, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v53, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v54, v8, v12, v10 bitop3:0xc8\n v_bitop3_b32 v55, v8, v12, v10 bitop3:0xc8\n"
#define BODY_OR_CONF \
    "v_or_b32 v40, v8, v12\n v_or_b32 v41, v8, v12\n v_or_b32 v42, v8, v12\n v_or_b32 v43, v8, v12\n" \
    "v_or_b32 v44, v8, v12\n v_or_b32 v45, v8, v12\n v_or_b32 v46, v8, v12\n v_or_b32 v47, v8, v12\n" \
    "v_or_b32 v48, v8, v12\n v_or_b32 v49, v8, v12\n v_or_b32 v50, v8, v12\n v_or_b32 v51, v8, v12\n" \
    "v_or_b32 v52, v8, v12\n v_or_b32 v53, v8, v12\n v_or_b32 v54, v8, v12\n v_or_b32 v55, v8, v12\n"
#define BODY_SUB_CONF \
    "v_sub_f32 v40, v8, v12\n v_sub_f32 v41, v8, v12\n v_sub_f32 v42, v8, v12\n v_sub_f32 v43, v8, v12\n" \
    "v_sub_f32 v44, v8, v12\n v_sub_f32 v45, v8, v12\n v_sub_f32 v46, v8, v12\n v_sub_f32 v47, v8, v12\n" \
    "v_sub_f32 v48, v8, v12\n v_sub_f32 v49, v8, v12\n v_sub_f32 v50, v8, v12\n v_sub_f32 v51, v8, v12\n" \
    "v_sub_f32 v52, v8, v12\n v_sub_f32 v53, v8, v12\n v_sub_f32 v54, v8, v12\n v_sub_f32 v55, v8, v12\n"
#define BODY_ADD3U \
    "v_add3_u32 v40, v8, v9, v10\n v_add3_u32 v41, v8, v9, v10\n v_add3_u32 v42, v8, v9, v10\n v_add3_u32 v43, v8, v9, v10\n" \
    "v_add3_u32 v44, v8, v9, v10\n v_add3_u32 v45, v8, v9, v10\n v_add3_u32 v46, v8, v9, v10\n v_add3_u32 v47, v8, v9, v10\n" \
    "v_add3_u32 v48, v8, v9, v10\n v_add3_u32 v49, v8, v9, v10\n v_add3_u32 v50, v8, v9, v10\n v_add3_u32 v51, v8, v9, v10\n" \
    "v_add3_u32 v52, v8, v9, v10\n v_add3_u32 v53, v8, v9, v10\n v_add3_u32 v54, v8, v9, v10\n v_add3_u32 v55, v8, v9, v10\n"
#define BODY_L_NEG0 \
    "v_fma_f32 v40, -v9, v12, v14\n v_fma_f32 v44, -v9, v12, v14\n v_fma_f32 v48, -v9, v12, v14\n v_fma_f32 v52, -v9, v12, v14\n" \
    "v_fma_f32 v56, -v9, v12, v14\n v_fma_f32 v60, -v9, v12, v14\n v_fma_f32 v64, -v9, v12, v14\n v_fma_f32 v68, -v9, v12, v14\n" \
    "v_fma_f32 v40, -v9, v12, v14\n v_fma_f32 v44, -v9, v12, v14\n v_fma_f32 v48, -v9, v12, v14\n v_fma_f32 v52, -v9, v12, v14\n" \
    "v_fma_f32 v56, -v9, v12, v14\n v_fma_f32 v60, -v9, v12, v14\n v_fma_f32 v64, -v9, v12, v14\n v_fma_f32 v68, -v9, v12, v14\n"
#define BODY_L_POS0 \
    "v_fma_f32 v40, v9, v12, v14\n v_fma_f32 v44, v9, v12, v14\n v_fma_f32 v48, v9, v12, v14\n v_fma_f32 v52, v9, v12, v14\n" \
    "v_fma_f32 v56, v9, v12, v14\n v_fma_f32 v60, v9, v12, v14\n v_fma_f32 v64, v9, v12, v14\n v_fma_f32 v68, v9, v12, v14\n" \
    "v_fma_f32 v40, v9, v12, v14\n v_fma_f32 v44, v9, v12, v14\n v_fma_f32 v48, v9, v12, v14\n v_fma_f32 v52, v9, v12, v14\n" \
    "v_fma_f32 v56, v9, v12, v14\n v_fma_f32 v60, v9, v12, v14\n v_fma_f32 v64, v9, v12, v14\n v_fma_f32 v68, v9, v12, v14\n"
#define BODY_L_NEG0_D3 \
    "v_fma_f32 v43, -v9, v12, v14\n v_fma_f32 v47, -v9, v12, v14\n v_fma_f32 v51, -v9, v12, v14\n v_fma_f32 v55, -v9, v12, v14\n" \
    "v_fma_f32 v59, -v9, v12, v14\n v_fma_f32 v63, -v9, v12, v14\n v_fma_f32 v67, -v9, v12, v14\n v_fma_f32 v71, -v9, v12, v14\n" \
    "v_fma_f32 v43, -v9, v12, v14\n v_fma_f32 v47, -v9, v12, v14\n v_fma_f32 v51, -v9, v12, v14\n v_fma_f32 v55, -v9, v12, v14\n" \
    "v_fma_f32 v59, -v9, v12, v14\n v_fma_f32 v63, -v9, v12, v14\n v_fma_f32 v67, -v9, v12, v14\n v_fma_f32 v71, -v9, v12, v14\n"
#define BODY_L_POS0_D3 \
    "v_fma_f32 v43, v9, v12, v14\n v_fma_f32 v47, v9, v12, v14\n v_fma_f32 v51, v9, v12, v14\n v_fma_f32 v55, v9, v12, v14\n" \
    "v_fma_f32 v59, v9, v12, v14\n v_fma_f32 v63, v9, v12, v14\n v_fma_f32 v67, v9, v12, v14\n v_fma_f32 v71, v9, v12, v14\n" \
    "v_fma_f32 v43, v9, v12, v14\n v_fma_f32 v47, v9, v12, v14\n v_fma_f32 v51, v9, v12, v14\n v_fma_f32 v55, v9, v12, v14\n" \
    "v_fma_f32 v59, v9, v12, v14\n v_fma_f32 v63, v9, v12, v14\n v_fma_f32 v67, v9, v12, v14\n v_fma_f32 v71, v9, v12, v14\n"
#define BODY_L_CHAIN \
    "v_fma_f32 v40, -v9, v12, v14\n v_fma_f32 v44, -v9, v12, v14\n v_fma_f32 v48, -v9, v12, v14\n v_fma_f32 v52, -v9, v12, v14\n" \
    "v_fma_f32 v56, -v9, v12, v14\n v_fma_f32 v60, -v9, v12, v14\n v_fma_f32 v64, -v9, v12, v14\n v_fma_f32 v68, -v9, v12, v14\n" \
    "v_fma_f32 v41, -v10, v13, v40\n v_fma_f32 v45, -v10, v13, v44\n v_fma_f32 v49, -v10, v13, v48\n v_fma_f32 v53, -v10, v13, v52\n" \
    "v_fma_f32 v57, -v10, v13, v56\n v_fma_f32 v61, -v10, v13, v60\n v_fma_f32 v65, -v10, v13, v64\n v_fma_f32 v69, -v10, v13, v68\n"
#define BODY_L_SQ \
    "v_fma_f32 v41, v40, v40, v15\n v_fma_f32 v45, v44, v44, v15\n v_fma_f32 v49, v48, v48, v15\n v_fma_f32 v53, v52, v52, v15\n" \
    "v_fma_f32 v57, v56, v56, v15\n v_fma_f32 v61, v60, v60, v15\n v_fma_f32 v65, v64, v64, v15\n v_fma_f32 v69, v68, v68, v15\n" \
    "v_fma_f32 v41, v40, v40, v15\n v_fma_f32 v45, v44, v44, v15\n v_fma_f32 v49, v48, v48, v15\n v_fma_f32 v53, v52, v52, v15\n" \
    "v_fma_f32 v57, v56, v56, v15\n v_fma_f32 v61, v60, v60, v15\n v_fma_f32 v65, v64, v64, v15\n v_fma_f32 v69, v68, v68, v15\n"

#define CLOBBERS "v8","v9","v10","v11","v12","v13","v14","v15","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","s20","s21","s22","s23","vcc"

#define KERNEL(NAME, BODY)                                                                   \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters)                       \
    {                                                                                        \
        asm volatile("v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 2.0\n v_mov_b32 v11, 4.0\n" \
                     "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 2.0\n v_mov_b32 v15, 4.0\n" \
                     "s_mov_b32 s20, 1.0\n s_mov_b32 s21, 0.5\n s_mov_b32 s22, 2.0\n s_mov_b32 s23, 4.0\n" ::: CLOBBERS); \
        for (int i = 0; i < iters; i++) asm volatile(REP16(BODY) ::: CLOBBERS);              \
        if (iters < 0) out[threadIdx.x] = 1.0f;                                              \
    }

KERNEL(k_mul_vv, BODY_MUL_VV)
KERNEL(k_mul_sv, BODY_MUL_SV)
KERNEL(k_add_vv, BODY_ADD_VV)
KERNEL(k_fma_vvv, BODY_FMA_VVV)
KERNEL(k_pkmul_vv, BODY_PKMUL_VV)
KERNEL(k_pkmul_vs, BODY_PKMUL_VS)
KERNEL(k_pkfma_vvv, BODY_PKFMA_VVV)
KERNEL(k_cmp, BODY_CMP)
KERNEL(k_max3, BODY_MAX3)

KERNEL(k_body_min_vv, BODY_MIN_VV)
KERNEL(k_body_max_vv, BODY_MAX_VV)
KERNEL(k_body_sub_vv, BODY_SUB_VV)
KERNEL(k_body_min_neg, BODY_MIN_NEG)
KERNEL(k_body_mul_abs, BODY_MUL_ABS)
KERNEL(k_body_fma_neg, BODY_FMA_NEG)
KERNEL(k_body_fma_s, BODY_FMA_S)
KERNEL(k_body_fma_k, BODY_FMA_K)
KERNEL(k_body_and, BODY_AND)
KERNEL(k_body_or, BODY_OR)
KERNEL(k_body_cndmask, BODY_CNDMASK)
KERNEL(k_body_mov, BODY_MOV)
KERNEL(k_body_subrev_s, BODY_SUBREV_S)
KERNEL(k_body_mac, BODY_MAC)
KERNEL(k_fma_2bank, BODY_FMA_2BANK)
KERNEL(k_fma_2bankb, BODY_FMA_2BANKB)
KERNEL(k_fma_noconf, BODY_FMA_NOCONF)
KERNEL(k_mul_2bank, BODY_MUL_2BANK)
KERNEL(k_fmac_s01, BODY_FMAC_S01)
KERNEL(k_fmac_d0, BODY_FMAC_D0)
KERNEL(k_fmac_d0i, BODY_FMAC_D0i)
KERNEL(k_fmac_d1i, BODY_FMAC_D1i)
KERNEL(k_fmac_ok, BODY_FMAC_OK)
KERNEL(k_fma_sq, BODY_FMA_SQ)
KERNEL(k_fma_sq_conf, BODY_FMA_SQ_CONF)
KERNEL(k_bitop3, BODY_BITOP3)
KERNEL(k_bitop3_conf, BODY_BITOP3_CONF)
KERNEL(k_or_conf, BODY_OR_CONF)
KERNEL(k_sub_conf, BODY_SUB_CONF)
KERNEL(k_add3u, BODY_ADD3U)
KERNEL(k_l_neg0, BODY_L_NEG0)
KERNEL(k_l_pos0, BODY_L_POS0)
KERNEL(k_l_neg0_d3, BODY_L_NEG0_D3)
KERNEL(k_l_pos0_d3, BODY_L_POS0_D3)
KERNEL(k_l_chain, BODY_L_CHAIN)
KERNEL(k_l_sq, BODY_L_SQ)

typedef void (*kern_t)(float *, int);

double run(kern_t k, int blocks, int iters, float *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    float *d;
    (void)hipMalloc(&d, 4096);
    struct { const char *name; kern_t k; } ks[] = {
        {"v_mul_f32 v,v,v", k_mul_vv}, {"v_mul_f32 v,s,v", k_mul_sv}, {"v_add_f32 v,v,v", k_add_vv},
        {"v_fma_f32 v,v,v,v", k_fma_vvv}, {"v_pk_mul_f32 v,v,v", k_pkmul_vv},
        {"v_pk_mul_f32 v,v,s(bcast)", k_pkmul_vs}, {"v_pk_fma_f32 v,v,v,v", k_pkfma_vvv},
        {"v_cmp_ge_f32", k_cmp}, {"v_max3_f32", k_max3},
        {"v_fma_f32 v40, -v9, v12, v14 ...", k_l_neg0}, {"v_fma_f32 v40, v9, v12, v14 ...", k_l_pos0}, {"v_fma_f32 v43, -v9, v12, v14 ...", k_l_neg0_d3}, {"v_fma_f32 v43, v9, v12, v14 ...", k_l_pos0_d3}, {"v_fma_f32 v40, -v9, v12, v14 ...", k_l_chain}, {"v_fma_f32 v41, v40, v40, v15 ...", k_l_sq}, 
        {"v_fmac_f32 v40, v8, v12", k_fmac_s01}, {"v_fmac_f32 v40, v8, v9", k_fmac_d0}, {"v_fmac_f32 v40, v8, v9", k_fmac_d0i}, {"v_fmac_f32 v40, v9, v8", k_fmac_d1i}, {"v_fmac_f32 v40, v9, v10", k_fmac_ok}, {"v_fma_f32 v40, v9, v9, v10", k_fma_sq}, {"v_fma_f32 v40, v9, v9, v13", k_fma_sq_conf}, {"v_bitop3_b32 v40, v8, v9, v10 bitop3:0xc8", k_bitop3}, {"v_bitop3_b32 v40, v8, v12, v10 bitop3:0xc8", k_bitop3_conf}, {"v_or_b32 v40, v8, v12", k_or_conf}, {"v_sub_f32 v40, v8, v12", k_sub_conf}, {"v_add3_u32 v40, v8, v9, v10", k_add3u}, 
        {"v_fma src0,src1 same bank", k_fma_2bank}, {"v_fma src1,src2 same bank", k_fma_2bankb}, {"v_fma 3 banks", k_fma_noconf}, {"v_mul src0,src1 same bank", k_mul_2bank},
        {"v_min_f32 vD, vA, vB", k_body_min_vv}, {"v_max_f32 vD, vA, vB", k_body_max_vv}, {"v_sub_f32 vD, vA, vB", k_body_sub_vv}, {"v_min_f32_e64 vD, vA, -vB", k_body_min_neg}, {"v_mul_f32_e64 vD, vA, -|vB|", k_body_mul_abs}, {"v_fma_f32 vD, vA, vB, -vC", k_body_fma_neg}, {"v_fma_f32 vD, vA, s20, vC", k_body_fma_s}, {"v_fma_f32 vD, vA, vB, 1.0", k_body_fma_k}, {"v_and_b32 vD, vA, vB", k_body_and}, {"v_or_b32 vD, vA, vB", k_body_or}, {"v_cndmask_b32 vD, vA, vB, vcc", k_body_cndmask}, {"v_mov_b32 vD, vA", k_body_mov}, {"v_subrev_f32 vD, s21, vB", k_body_subrev_s}, {"v_fmac_f32 vD, vA, vB", k_body_mac}};
    const int iters = 2000;
    for (int wps : {4}) {
        for (auto &e : ks) {
            const int blocks = p.multiProcessorCount * wps;
            const double ms = run(e.k, blocks, iters, d);
            const double instr_per_simd = (double)wps * iters * 256;  // 16*16 per trip per wave
            printf("waves/SIMD %d  %-28s %8.3f ms  %5.2f ns/instr/SIMD (= %.2f cycles @2.4 GHz)\n", wps, e.name,
                   ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
        }
    }
    return 0;
}
