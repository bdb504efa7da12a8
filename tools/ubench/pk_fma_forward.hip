// Micro-test (gfx950), r06 -- tools/asm_bisect narrowed the run-to-run differing bits of the SLP-vectorised build of csrc/emd.hip to ONE
// instruction of emd_mfma_cols_kernel<0>: replacing
//     v_pk_fma_f32 v[132:133], v[172:173], v[112:113], v[132:133] op_sel:[0,1,0]
// by the two v_fma_f32 it stands for makes the kernel repeat; draining every MFMA (48 wait states) does not.  Its neighbourhood:
//     s_waitcnt vmcnt(6)                              ; v[112:115] = a global_load_dwordx4 that has just landed
//     v_mov_b32 v140, v115
//     v_pk_fma_f32 v[132:133], v[170:171], v[112:113], v[132:133] op_sel_hi:[1,0,1]
//     s_nop 0
//     v_pk_fma_f32 v[132:133], v[172:173], v[112:113], v[132:133] op_sel:[0,1,0]      <- the one
//     v_mfma_f32_32x32x16_f16 v[34:49], v[142:145], v[78:81], v[34:49]
//     v_fma_f32 v132, v174, v114, v132                ; reads the packed result one instruction (an MFMA) later
//     v_fma_f32 v133, v175, v114, v133
//     v_fma_f32 v132, v176, v140, v132
//     v_fma_f32 v133, v177, v140, v133
// This file replays that neighbourhood in isolation, in forms that take pieces away, and compares v[132:133] with the same
// arithmetic done by fmaf() on values read back from memory (what the packed instructions stand for).
//   build: hipcc --offload-arch=gfx950 -O3 pk_fma_forward.hip -o pk_fma_forward ; run: ./pk_fma_forward
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

#define PRE                                                                                                                     \
    "v_mov_b32 v170, %[e0]\n v_mov_b32 v171, %[e1]\n v_mov_b32 v172, %[e2]\n v_mov_b32 v173, %[e3]\n"                           \
    "v_mov_b32 v174, %[e4]\n v_mov_b32 v175, %[e5]\n v_mov_b32 v176, %[e6]\n v_mov_b32 v177, %[e7]\n"                           \
    "v_mov_b32 v132, %[a0]\n v_mov_b32 v133, %[a1]\n"                                                                           \
    "v_mov_b32 v142, %[fa0]\n v_mov_b32 v143, %[fa1]\n v_mov_b32 v144, %[fa2]\n v_mov_b32 v145, %[fa3]\n"                       \
    "v_mov_b32 v78, %[fb0]\n v_mov_b32 v79, %[fb1]\n v_mov_b32 v80, %[fb2]\n v_mov_b32 v81, %[fb3]\n"                           \
    "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"        \
    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n"        \
    "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n"                                              \
    "v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"                                          \
    "global_load_dwordx4 v[112:115], %[pw], off\n"                                                                              \
    "s_waitcnt vmcnt(0)\n"
// ... the same with six more loads in flight BEHIND the weights' (the kernel's next-tile prefetch: s_waitcnt vmcnt(6) lets the
// sequence run while they land), issued %[dl] x 16 cycles before the sequence so that the landing sweeps across it
#define PRE6                                                                                                                    \
    "v_mov_b32 v170, %[e0]\n v_mov_b32 v171, %[e1]\n v_mov_b32 v172, %[e2]\n v_mov_b32 v173, %[e3]\n"                           \
    "v_mov_b32 v174, %[e4]\n v_mov_b32 v175, %[e5]\n v_mov_b32 v176, %[e6]\n v_mov_b32 v177, %[e7]\n"                           \
    "v_mov_b32 v132, %[a0]\n v_mov_b32 v133, %[a1]\n"                                                                           \
    "v_mov_b32 v142, %[fa0]\n v_mov_b32 v143, %[fa1]\n v_mov_b32 v144, %[fa2]\n v_mov_b32 v145, %[fa3]\n"                       \
    "v_mov_b32 v78, %[fb0]\n v_mov_b32 v79, %[fb1]\n v_mov_b32 v80, %[fb2]\n v_mov_b32 v81, %[fb3]\n"                           \
    "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"        \
    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n"        \
    "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n"                                              \
    "v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"                                          \
    "global_load_dwordx4 v[112:115], %[pw], off\n"                                                                              \
    "global_load_dwordx4 v[108:111], %[px], off\n global_load_dwordx4 v[104:107], %[px], off offset:32\n"                       \
    "global_load_dwordx4 v[100:103], %[px], off offset:64\n global_load_dwordx4 v[94:97], %[px], off offset:96\n"              \
    "global_load_dwordx4 v[90:93], %[px], off offset:128\n global_load_dwordx4 v[74:77], %[px], off offset:160\n"              \
    "s_mov_b32 s20, %[dl]\n"                                                                                                    \
    "1:\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 2f\n s_nop 11\n s_sub_u32 s20, s20, 1\n s_branch 1b\n 2:\n"                    \
    "s_waitcnt vmcnt(6)\n"
#define PRE6X                                                                                                                   \
    "v_mov_b32 v170, %[e0]\n v_mov_b32 v171, %[e1]\n v_mov_b32 v172, %[e2]\n v_mov_b32 v173, %[e3]\n"                           \
    "v_mov_b32 v174, %[e4]\n v_mov_b32 v175, %[e5]\n v_mov_b32 v176, %[e6]\n v_mov_b32 v177, %[e7]\n"                           \
    "v_mov_b32 v132, %[a0]\n v_mov_b32 v133, %[a1]\n"                                                                           \
    "v_mov_b32 v142, %[fa0]\n v_mov_b32 v143, %[fa1]\n v_mov_b32 v144, %[fa2]\n v_mov_b32 v145, %[fa3]\n"                       \
    "v_mov_b32 v78, %[fb0]\n v_mov_b32 v79, %[fb1]\n v_mov_b32 v80, %[fb2]\n v_mov_b32 v81, %[fb3]\n"                           \
    "v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"        \
    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n"        \
    "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n"                                              \
    "v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"                                          \
    "global_load_dwordx4 v[112:115], %[pw], off\n"                                                                              \
    "global_load_dwordx4 v[108:111], %[px], off\n global_load_dwordx4 v[104:107], %[px], off offset:32\n"                       \
    "global_load_dwordx4 v[100:103], %[px], off offset:64\n global_load_dwordx4 v[94:97], %[px], off offset:96\n"              \
    "global_load_dwordx4 v[90:93], %[px], off offset:128\n global_load_dwordx4 v[74:77], %[px], off offset:160\n"              \
    "s_mov_b32 s20, %[dl]\n"                                                                                                    \
    "1:\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 2f\n s_nop 11\n s_sub_u32 s20, s20, 1\n s_branch 1b\n 2:\n"                    \
    ""
#define DRAIN6 "s_waitcnt vmcnt(0)\n"
// the kernel's instructions in front of the neighbourhood, as they stand in its text: three transcendentals, a packed fma that
// is itself followed by an MFMA (whose destination overlaps the packed fma's third source), the scalar terms of r = 11
#define LEAD                                                                                                                    \
    "v_mov_b32 v31, %[e1]\n v_mov_b32 v32, %[e3]\n v_mov_b32 v33, %[e5]\n v_mov_b32 v18, %[a0]\n v_mov_b32 v19, %[a1]\n"               \
    "v_mov_b32 v166, 0\n v_mov_b32 v167, 0\n v_mov_b32 v168, 0\n v_mov_b32 v169, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n" \
    "v_mov_b32 v138, %[fa0]\n v_mov_b32 v139, %[fa1]\n v_mov_b32 v140, %[fa2]\n v_mov_b32 v141, %[fa3]\n"                       \
    "v_mov_b32 v82, %[fb0]\n v_mov_b32 v83, %[fb1]\n v_mov_b32 v84, %[fb2]\n v_mov_b32 v85, %[fb3]\n"                           \
    "v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n" \
    "v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n" \
    "s_nop 7\n"                                                                                                                 \
    "v_exp_f32 v178, v31\n v_exp_f32 v179, v32\n v_exp_f32 v180, v33\n"                                                        \
    "v_pk_fma_f32 v[132:133], v[166:167], v[118:119], v[18:19] op_sel_hi:[1,0,1]\n"                                              \
    "v_mfma_f32_32x32x16_f16 v[18:33], v[138:141], v[82:85], v[2:17]\n"                                                         \
    "v_mov_b32 v138, v119\n"                                                                                                    \
    "v_fma_f32 v132, v168, v138, v132\n v_fma_f32 v133, v169, v138, v133\n"
// the kernel's text from its line "v_exp_f32 v163, v26" to the neighbourhood: ten transcendentals interleaved with packed fmas (the
// transcendental unit is kept busy; v171 / v173 / v175 / v177 -- the HIGH halves of the neighbourhood's packed sources -- are
// results still on their way), the weights of every earlier term are zero so that the accumulators pass through unchanged
#define LEAD2                                                                                                                   \
    "v_mov_b32 v26, %[x0]\n v_mov_b32 v27, %[x1]\n v_mov_b32 v28, %[x2]\n v_mov_b32 v29, %[x3]\n"                               \
    "v_mov_b32 v30, %[x4]\n v_mov_b32 v31, %[x5]\n v_mov_b32 v32, %[x6]\n v_mov_b32 v33, %[x7]\n"                               \
    "v_mov_b32 v18, %[a0]\n v_mov_b32 v19, %[a1]\n"                                                                             \
    "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n"   \
    "v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v149, 0\n"                                                              \
    "v_mov_b32 v154, %[e0]\n v_mov_b32 v155, %[e1]\n v_mov_b32 v156, %[e2]\n v_mov_b32 v157, %[e3]\n v_mov_b32 v158, %[e4]\n"    \
    "v_mov_b32 v159, %[e5]\n v_mov_b32 v160, %[e6]\n v_mov_b32 v161, %[e7]\n v_mov_b32 v162, %[e0]\n v_mov_b32 v164, %[e2]\n"    \
    "v_mov_b32 v166, %[e4]\n v_mov_b32 v168, %[e6]\n"                                                                           \
    "v_mov_b32 v138, %[fa0]\n v_mov_b32 v139, %[fa1]\n v_mov_b32 v140, %[fa2]\n v_mov_b32 v141, %[fa3]\n"                       \
    "v_mov_b32 v82, %[fb0]\n v_mov_b32 v83, %[fb1]\n v_mov_b32 v84, %[fb2]\n v_mov_b32 v85, %[fb3]\n"                           \
    "v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n" \
    "v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n" \
    "s_nop 7\n"                                                                                                                 \
    "v_exp_f32 v163, v26\n"                                                                                                     \
    "v_pk_fma_f32 v[18:19], v[154:155], v[120:121], v[18:19] op_sel_hi:[1,0,1]\n"                                                \
    "v_exp_f32 v165, v27\n"                                                                                                     \
    "v_pk_fma_f32 v[18:19], v[156:157], v[120:121], v[18:19] op_sel:[0,1,0]\n"                                                   \
    "v_exp_f32 v167, v28\n"                                                                                                     \
    "v_pk_fma_f32 v[18:19], v[158:159], v[122:123], v[18:19] op_sel_hi:[1,0,1]\n"                                                \
    "v_mov_b32 v148, v123\n"                                                                                                    \
    "v_pk_fma_f32 v[18:19], v[160:161], v[148:149], v[18:19] op_sel_hi:[1,0,1]\n"                                                \
    "v_exp_f32 v169, v29\n"                                                                                                     \
    "v_pk_fma_f32 v[18:19], v[162:163], v[116:117], v[18:19] op_sel_hi:[1,0,1]\n"                                                \
    "v_exp_f32 v171, v30\n"                                                                                                     \
    "v_pk_fma_f32 v[18:19], v[164:165], v[116:117], v[18:19] op_sel:[0,1,0]\n"                                                   \
    "v_exp_f32 v173, v31\n"                                                                                                     \
    "v_exp_f32 v175, v32\n"                                                                                                     \
    "v_exp_f32 v177, v33\n"                                                                                                     \
    "v_pk_fma_f32 v[132:133], v[166:167], v[118:119], v[18:19] op_sel_hi:[1,0,1]\n"                                              \
    "v_mfma_f32_32x32x16_f16 v[18:33], v[138:141], v[82:85], v[2:17]\n"                                                         \
    "v_mov_b32 v138, v119\n"                                                                                                    \
    "v_fma_f32 v132, v168, v138, v132\n v_fma_f32 v133, v169, v138, v133\n"
// (PRE6X sets v170..v177 first; LEAD2 then overwrites the odd ones with transcendental results)
#define OPSL2                                                                                                                   \
    : [r0] "=&v"(r0), [r1] "=&v"(r1), [m0] "=&v"(m0)                                                                            \
    : [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]), [e4] "v"(e[4]), [e5] "v"(e[5]), [e6] "v"(e[6]),          \
      [e7] "v"(e[7]), [a0] "v"(a0), [a1] "v"(a1), [fa0] "v"(fa.x), [fa1] "v"(fa.y), [fa2] "v"(fa.z), [fa3] "v"(fa.w),          \
      [fb0] "v"(fb.x), [fb1] "v"(fb.y), [fb2] "v"(fb.z), [fb3] "v"(fb.w), [pw] "v"(pw), [px] "v"(px), [dl] "s"(dl),            \
      [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]) \
    : "memory", "scc", "s20", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", \
      "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",   \
      "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",                                 \
      "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v90", "v91", "v92", "v93", "v94", "v95",  \
      "v96", "v97", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",  \
      "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v132", "v133", "v138", "v139", "v140", "v141", \
      "v142", "v143", "v144", "v145", "v148", "v149", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", \
      "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177"
#define OPSL                                                                                                                    \
    : [r0] "=&v"(r0), [r1] "=&v"(r1), [m0] "=&v"(m0)                                                                            \
    : [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]), [e4] "v"(e[4]), [e5] "v"(e[5]), [e6] "v"(e[6]),          \
      [e7] "v"(e[7]), [a0] "v"(a0), [a1] "v"(a1), [fa0] "v"(fa.x), [fa1] "v"(fa.y), [fa2] "v"(fa.z), [fa3] "v"(fa.w),          \
      [fb0] "v"(fb.x), [fb1] "v"(fb.y), [fb2] "v"(fb.z), [fb3] "v"(fb.w), [pw] "v"(pw), [px] "v"(px), [dl] "s"(dl)             \
    : "memory", "scc", "s20", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", \
      "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36",   \
      "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",                                 \
      "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v90", "v91", "v92", "v93", "v94", "v95",  \
      "v96", "v97", "v100", "v101",  \
      "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v118", "v119", "v132",    \
      "v133", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", \
      "v178", "v179", "v180"
#define OPS6                                                                                                                    \
    : [r0] "=&v"(r0), [r1] "=&v"(r1), [m0] "=&v"(m0)                                                                            \
    : [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]), [e4] "v"(e[4]), [e5] "v"(e[5]), [e6] "v"(e[6]),          \
      [e7] "v"(e[7]), [a0] "v"(a0), [a1] "v"(a1), [fa0] "v"(fa.x), [fa1] "v"(fa.y), [fa2] "v"(fa.z), [fa3] "v"(fa.w),          \
      [fb0] "v"(fb.x), [fb1] "v"(fb.y), [fb2] "v"(fb.z), [fb3] "v"(fb.w), [pw] "v"(pw), [px] "v"(px), [dl] "s"(dl)             \
    : "memory", "scc", "s20", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
      "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v100", "v101",  \
      "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v132",    \
      "v133", "v140", "v142", "v143", "v144", "v145", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177"
#define MOV140 "v_mov_b32 v140, v115\n"
#define PK43 "v_pk_fma_f32 v[132:133], v[170:171], v[112:113], v[132:133] op_sel_hi:[1,0,1]\n"
#define PK44 "v_pk_fma_f32 v[132:133], v[172:173], v[112:113], v[132:133] op_sel:[0,1,0]\n"
#define SC44 "v_fma_f32 v132, v172, v113, v132\n v_fma_f32 v133, v173, v113, v133\n"
#define MF "v_mfma_f32_32x32x16_f16 v[34:49], v[142:145], v[78:81], v[34:49]\n"
#define TAIL                                                                                                                    \
    "v_fma_f32 v132, v174, v114, v132\n v_fma_f32 v133, v175, v114, v133\n"                                                     \
    "v_fma_f32 v132, v176, v140, v132\n v_fma_f32 v133, v177, v140, v133\n"                                                     \
    "s_nop 15\n s_nop 15\n s_nop 15\n"                                                                                          \
    "v_mov_b32 %[r0], v132\n v_mov_b32 %[r1], v133\n v_mov_b32 %[m0], v34\n"
#define OPS                                                                                                                     \
    : [r0] "=&v"(r0), [r1] "=&v"(r1), [m0] "=&v"(m0)                                                                            \
    : [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]), [e4] "v"(e[4]), [e5] "v"(e[5]), [e6] "v"(e[6]),          \
      [e7] "v"(e[7]), [a0] "v"(a0), [a1] "v"(a1), [fa0] "v"(fa.x), [fa1] "v"(fa.y), [fa2] "v"(fa.z), [fa3] "v"(fa.w),          \
      [fb0] "v"(fb.x), [fb1] "v"(fb.y), [fb2] "v"(fb.z), [fb3] "v"(fb.w), [pw] "v"(pw)                                         \
    : "memory", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
      "v78", "v79", "v80", "v81", "v112", "v113", "v114", "v115", "v132", "v133", "v140", "v142", "v143", "v144", "v145",      \
      "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177"

static const char *FORM_NAME[] = {
    "0 the neighbourhood as the compiler left it",
    "1 without the MFMA",
    "2 s_nop 1 between the packed fma and the MFMA",
    "3 s_nop 1 between the MFMA and the reader",
    "4 the packed fma replaced by its two v_fma_f32 (control)",
    "5 without the first packed fma and its s_nop",
    "6 the loaded weights settled (s_nop 15 behind the waitcnt)",
    "7 s_nop 3 in front of the packed fma",
    "8 form 0 with six more loads landing around the sequence",
    "9 form 8 with s_nop 1 between the packed fma and the MFMA",
    "10 form 8 with the packed fma as two v_fma_f32 (control)",
    "11 form 8 without the MFMA",
    "12 form 8 behind the kernel's own lead-in (exps, packed fma + MFMA)",
    "13 form 12 with s_nop 1 between the packed fma and the MFMA",
    "14 form 12 with the packed fma as two v_fma_f32 (control)",
    "15 form 12, the six loads from the same resident lines (a burst)",
    "16 form 15 with s_nop 1 between the packed fma and the MFMA",
    "17 form 15 with the packed fma as two v_fma_f32 (control)",
    "18 the kernel's text from 26 instructions before: the packed sources' high halves are transcendental results on their way",
    "19 form 18 with s_nop 0 between the packed fma and the MFMA",
    "20 form 18 with the packed fma as two v_fma_f32 (control)",
};
constexpr int NFORM = 21;

template <int FORM>
__global__ __launch_bounds__(256) void probe(const f4 *__restrict__ W, const float *__restrict__ E, const u4 *__restrict__ F, int iters,
                                             int nset, unsigned long long *bad, const f4 *__restrict__ X, int xn) {
    const int lane = threadIdx.x & 63;
    unsigned long long b0 = 0, b1 = 0;
    for (int it = 0; it < iters; ++it) {
        const int set = (it * 5 + blockIdx.x + (threadIdx.x >> 6)) % nset;
        const f4 *pw = W + set * 64 + lane;
        float e[8];
        for (int u = 0; u < 8; ++u) e[u] = E[(set * 8 + u) * 64 + lane];
        const float a0 = E[((set + 1) % nset * 8) * 64 + lane], a1 = E[((set + 1) % nset * 8 + 1) * 64 + lane];
        const u4 fa = F[set * 64 + lane], fb = F[(nset + set) * 64 + lane];
        float r0, r1, m0;
        if constexpr (FORM == 0) asm volatile(PRE MOV140 PK43 "s_nop 0\n" PK44 MF TAIL OPS);
        if constexpr (FORM == 1) asm volatile(PRE MOV140 PK43 "s_nop 0\n" PK44 TAIL OPS);
        if constexpr (FORM == 2) asm volatile(PRE MOV140 PK43 "s_nop 0\n" PK44 "s_nop 1\n" MF TAIL OPS);
        if constexpr (FORM == 3) asm volatile(PRE MOV140 PK43 "s_nop 0\n" PK44 MF "s_nop 1\n" TAIL OPS);
        if constexpr (FORM == 4) asm volatile(PRE MOV140 PK43 "s_nop 0\n" SC44 MF TAIL OPS);
        if constexpr (FORM == 5) asm volatile(PRE MOV140 PK44 MF TAIL OPS);
        if constexpr (FORM == 6) asm volatile(PRE "s_nop 15\n" MOV140 PK43 "s_nop 0\n" PK44 MF TAIL OPS);
        if constexpr (FORM == 7) asm volatile(PRE MOV140 PK43 "s_nop 0\n s_nop 3\n" PK44 MF TAIL OPS);
        // (the extra loads walk a 64 MiB array: L2 hits, misses and everything between)
        const f4 *px = X + (((size_t)it * 2654435761u + blockIdx.x * 40503u + threadIdx.x * 12u) % (size_t)(xn - 16));
        const int dl = __builtin_amdgcn_readfirstlane((it * 7 + blockIdx.x) % 96);
        if constexpr (FORM == 8) asm volatile(PRE6 MOV140 PK43 "s_nop 0\n" PK44 MF TAIL DRAIN6 OPS6);
        if constexpr (FORM == 9) asm volatile(PRE6 MOV140 PK43 "s_nop 0\n" PK44 "s_nop 1\n" MF TAIL DRAIN6 OPS6);
        if constexpr (FORM == 10) asm volatile(PRE6 MOV140 PK43 "s_nop 0\n" SC44 MF TAIL DRAIN6 OPS6);
        if constexpr (FORM == 11) asm volatile(PRE6 MOV140 PK43 "s_nop 0\n" PK44 TAIL DRAIN6 OPS6);
        // (the lead-in sits between the loads' issue and the waitcnt, where the kernel has it; dl = 0..3 here)
        if constexpr (FORM == 12) asm volatile(PRE6X LEAD "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" PK44 MF TAIL DRAIN6 OPSL);
        if constexpr (FORM == 13) asm volatile(PRE6X LEAD "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" PK44 "s_nop 1\n" MF TAIL DRAIN6 OPSL);
        if constexpr (FORM == 14) asm volatile(PRE6X LEAD "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" SC44 MF TAIL DRAIN6 OPSL);
        if constexpr (FORM >= 18) {
            const f4 *px = (const f4 *)((const char *)(W + ((set + 3) % nset) * 64 + lane));
            const int dl = 0;
            float x[8];
            for (int u = 0; u < 8; ++u) x[u] = -30.0f * E[((set + 2) % nset * 8 + u) * 64 + lane] - 0.25f * u;      // exp2 arguments in (-32, 0]
            if constexpr (FORM == 18) asm volatile(PRE6X LEAD2 "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" PK44 MF TAIL DRAIN6 OPSL2);
            if constexpr (FORM == 19) asm volatile(PRE6X LEAD2 "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" PK44 "s_nop 0\n" MF TAIL DRAIN6 OPSL2);
            if constexpr (FORM == 20) asm volatile(PRE6X LEAD2 "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" SC44 MF TAIL DRAIN6 OPSL2);
            // the odd sources are the transcendental results
            e[1] = __builtin_amdgcn_exp2f(x[4]); e[3] = __builtin_amdgcn_exp2f(x[5]); e[5] = __builtin_amdgcn_exp2f(x[6]); e[7] = __builtin_amdgcn_exp2f(x[7]);
        } else if constexpr (FORM >= 15) {       // all seven loads from neighbouring resident lines, no delay: the wave resumes as the burst lands
            const f4 *px = (const f4 *)((const char *)(W + ((set + 3) % nset) * 64 + lane));
            const int dl = 0;
            if constexpr (FORM == 15) asm volatile(PRE6X LEAD "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" PK44 MF TAIL DRAIN6 OPSL);
            if constexpr (FORM == 16) asm volatile(PRE6X LEAD "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" PK44 "s_nop 1\n" MF TAIL DRAIN6 OPSL);
            if constexpr (FORM == 17) asm volatile(PRE6X LEAD "s_waitcnt vmcnt(6)\n" MOV140 PK43 "s_nop 0\n" SC44 MF TAIL DRAIN6 OPSL);
        }
        const f4 w = *pw;
        float x0 = a0, x1 = a1;
        if (FORM != 5) { x0 = __builtin_fmaf(e[0], w.x, x0); x1 = __builtin_fmaf(e[1], w.x, x1); }
        x0 = __builtin_fmaf(e[2], w.y, x0); x1 = __builtin_fmaf(e[3], w.y, x1);
        x0 = __builtin_fmaf(e[4], w.z, x0); x1 = __builtin_fmaf(e[5], w.z, x1);
        x0 = __builtin_fmaf(e[6], w.w, x0); x1 = __builtin_fmaf(e[7], w.w, x1);
        b0 += __float_as_uint(x0) != __float_as_uint(r0);
        b1 += __float_as_uint(x1) != __float_as_uint(r1);
        if (m0 != m0) b0 += 1ull << 40;          // (keeps the MFMA's result alive)
    }
    if (b0) atomicAdd(bad, b0);
    if (b1) atomicAdd(bad + 1, b1);
}

template <int FORM>
void run(const f4 *W, const float *E, const u4 *F, int nset, unsigned long long *bad, int waves_per_simd, const f4 *X, int xn) {
    hipMemset(bad, 0, 16);
    const int iters = 4000, blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<FORM>), dim3(blocks), dim3(256), 0, 0, W, E, F, iters, nset, bad, X, xn);
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("form %-62s %d wave(s)/SIMD: %.3g lane-iterations, wrong low results %llu, wrong high results %llu\n", FORM_NAME[FORM],
           waves_per_simd, (double)blocks * 256 * iters, h[0], h[1]);
}

// the kernel's situation: ONE wave per workgroup, two workgroups in the grid, the sequence executed ONCE per launch (cold
// instruction fetch, idle chip), a few thousand launches with other kernels between them
template <int FORM>
void run_cold(const f4 *W, const float *E, const u4 *F, int nset, unsigned long long *bad, const f4 *X, int xn, int launches, int blocks) {
    hipMemset(bad, 0, 16);
    for (int l = 0; l < launches; ++l) {
        hipLaunchKernelGGL((probe<FORM>), dim3(blocks), dim3(64), 0, 0, W, E, F, 1, nset, bad, X, xn);
        if (l % 3 == 0) hipMemsetAsync((void *)X, 0, 4096, 0);         // (another kernel between two launches, as in a real stream)
    }
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("cold, %d launches of %d single-wave workgroup(s), form %-58s wrong low results %llu, wrong high results %llu\n", launches, blocks,
           FORM_NAME[FORM], h[0], h[1]);
}

int main(int argc, char **argv) {
    // ./pk_fma_forward denorm : one value in eight of the sources / accumulators is a DENORMAL fp32 number (the kernel's accumulators
    // hold such sums: exp2 results down to 2^-126 times weights below 1)
    const bool denorm = argc > 1 && argv[1][0] == 'd';
    printf("sources %s\n", denorm ? "with denormals" : "normal numbers and zeros only");
    const int nset = 8;
    f4 *W; float *E; u4 *F; unsigned long long *bad;
    hipMalloc(&W, (nset + 1) * 64 * sizeof(f4)); hipMemset(W, 0, (nset + 1) * 64 * sizeof(f4)); hipMalloc(&E, nset * 8 * 64 * sizeof(float)); hipMalloc(&F, 2 * nset * 64 * sizeof(u4));
    hipMalloc(&bad, 16);
    const int xn = 1 << 22;                            // 4 Mi float4 = 64 MiB
    f4 *X; hipMalloc(&X, (size_t)xn * sizeof(f4)); hipMemset(X, 0, (size_t)xn * sizeof(f4));
    srand(11);
    float *hw = (float *)malloc(nset * 64 * 16), *he = (float *)malloc(nset * 8 * 64 * 4);
    uint32_t *hf = (uint32_t *)malloc(2 * nset * 64 * 16);
    for (int i = 0; i < nset * 64 * 4; ++i) hw[i] = (rand() % 4 == 0) ? 0.f : 0.5f + rand() / (float)RAND_MAX * 1.5f;
    for (int i = 0; i < nset * 8 * 64; ++i) {
        he[i] = (rand() % 3 == 0) ? 0.f : ldexpf(0.5f + rand() / (float)RAND_MAX, -(rand() % 30));
        if (denorm && rand() % 8 == 0) he[i] = ldexpf(0.5f + rand() / (float)RAND_MAX, -128 - rand() % 18);
    }
    for (int i = 0; i < 2 * nset * 64 * 4; ++i) {
        uint32_t w = 0;
        for (int hh = 0; hh < 2; ++hh) w |= ((uint32_t)((rand() & 1) << 15) | (uint32_t)((13 + rand() % 3) << 10) | (rand() & 0x3ff)) << (16 * hh);
        hf[i] = w;
    }
    hipMemcpy(W, hw, nset * 64 * 16, hipMemcpyHostToDevice); hipMemcpy(E, he, nset * 8 * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(F, hf, 2 * nset * 64 * 16, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; w *= 2) {
        run<0>(W, E, F, nset, bad, w, X, xn); run<1>(W, E, F, nset, bad, w, X, xn); run<2>(W, E, F, nset, bad, w, X, xn);
        run<3>(W, E, F, nset, bad, w, X, xn); run<4>(W, E, F, nset, bad, w, X, xn); run<5>(W, E, F, nset, bad, w, X, xn);
        run<6>(W, E, F, nset, bad, w, X, xn); run<7>(W, E, F, nset, bad, w, X, xn); run<8>(W, E, F, nset, bad, w, X, xn);
        run<9>(W, E, F, nset, bad, w, X, xn); run<10>(W, E, F, nset, bad, w, X, xn); run<11>(W, E, F, nset, bad, w, X, xn);
        run<12>(W, E, F, nset, bad, w, X, xn); run<13>(W, E, F, nset, bad, w, X, xn); run<14>(W, E, F, nset, bad, w, X, xn);
        run<15>(W, E, F, nset, bad, w, X, xn); run<16>(W, E, F, nset, bad, w, X, xn); run<17>(W, E, F, nset, bad, w, X, xn);
        run<18>(W, E, F, nset, bad, w, X, xn); run<19>(W, E, F, nset, bad, w, X, xn); run<20>(W, E, F, nset, bad, w, X, xn);
    }
    for (int blocks = 2; blocks <= 32; blocks *= 4) {
        run_cold<0>(W, E, F, nset, bad, X, xn, 4000, blocks); run_cold<2>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<4>(W, E, F, nset, bad, X, xn, 4000, blocks); run_cold<8>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<9>(W, E, F, nset, bad, X, xn, 4000, blocks); run_cold<10>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<12>(W, E, F, nset, bad, X, xn, 4000, blocks); run_cold<13>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<14>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<15>(W, E, F, nset, bad, X, xn, 4000, blocks); run_cold<16>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<17>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<18>(W, E, F, nset, bad, X, xn, 4000, blocks); run_cold<19>(W, E, F, nset, bad, X, xn, 4000, blocks);
        run_cold<20>(W, E, F, nset, bad, X, xn, 4000, blocks);
    }
    return 0;
}
