// Issue-rate micro-benchmarks for ONE wavefront per SIMD on gfx950 (developer tool).
// Every test is one asm block: init, s_memtime, a counted loop over a .rept'ed body,
// s_memtime.  Prints shader cycles per body and per instruction.  The kernel claims 512
// registers (v255 / a255 clobbered) and 40 KB of LDS, so four workgroups = four wavefronts
// share a CU, one per SIMD, like k_sweep_reg<96,3>.
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

extern __shared__ __attribute__((aligned(16))) double lds[];

#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
  "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133", \
  "v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151", \
  "v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169", \
  "v255","a0","a1","a2","a3","a4","a5","a6","a7","a255","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","vcc","scc","memory"

// v[100:101] .. v[130:131]: doubles ~1.0 ; v140: lane*8 (LDS address) ; v141: lane*16 ; v142: lane*32
#define INIT \
  "v_mov_b32 v100, 0\n v_mov_b32 v101, 0x3ff00000\n" \
  "v_mov_b32 v102, 0\n v_mov_b32 v103, 0x3fd00000\n" \
  "v_mov_b32 v104, 0\n v_mov_b32 v105, 0x3ff00000\n" \
  "v_mov_b32 v106, 0\n v_mov_b32 v107, 0x3ff00000\n" \
  "v_mov_b32 v108, 0\n v_mov_b32 v109, 0x3ff00000\n" \
  "v_mov_b32 v110, 0\n v_mov_b32 v111, 0x3ff00000\n" \
  "v_mov_b32 v112, 0\n v_mov_b32 v113, 0x3ff00000\n" \
  "v_mov_b32 v114, 0\n v_mov_b32 v115, 0x3ff00000\n" \
  "v_mov_b32 v116, 0\n v_mov_b32 v117, 0x3ff00000\n" \
  "v_mov_b32 v118, 0\n v_mov_b32 v119, 0x3ff00000\n" \
  "v_mov_b32 v120, 0\n v_mov_b32 v121, 0x3ff00000\n" \
  "v_mov_b32 v122, 0\n v_mov_b32 v123, 0x3ff00000\n" \
  "v_mov_b32 v124, 0\n v_mov_b32 v125, 0x3ff00000\n" \
  "v_mov_b32 v126, 0\n v_mov_b32 v127, 0x3ff00000\n" \
  "v_mov_b32 v128, 0\n v_mov_b32 v129, 0x3ff00000\n" \
  "v_mov_b32 v130, 0\n v_mov_b32 v131, 0x3ff00000\n" \
  "v_mbcnt_lo_u32_b32 v140, -1, 0\n v_mbcnt_hi_u32_b32 v140, -1, v140\n" \
  "v_lshlrev_b32 v141, 4, v140\n v_lshlrev_b32 v142, 5, v140\n v_lshlrev_b32 v140, 3, v140\n" \
  "v_mov_b32 v143, 0\n"

#define TEST(NAME, NINSTR, BODY)                                                                   \
  __global__ void __launch_bounds__(64) NAME(long long *out, int iters, const char *gbuf) {                          \
    unsigned lo, hi;                                                                               \
    if (threadIdx.x < 64) for (int i = threadIdx.x; i < 5000; i += 64) lds[i] = 1.0;               \
    __syncthreads();                                                                               \
    asm volatile(INIT                                                                              \
                 "s_mov_b32 s22, %2\n s_mov_b32 s26, %3\n s_mov_b32 s27, %4\n"                     \
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                 \
                 "s_memtime s[20:21]\n s_waitcnt lgkmcnt(0)\n"                                     \
                 "1:\n .rept 16\n" BODY ".endr\n"                                                  \
                 "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n"               \
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                 \
                 "s_memtime s[24:25]\n s_waitcnt lgkmcnt(0)\n"                                     \
                 "s_sub_u32 %0, s24, s20\n s_subb_u32 %1, s25, s21\n"                              \
                 : "=s"(lo), "=s"(hi) : "s"(iters), "s"((unsigned)(size_t)gbuf), "s"((unsigned)((size_t)gbuf >> 32)) : CLOB);                                        \
    if (threadIdx.x == 0) out[blockIdx.x] = ((long long)hi << 32) | lo;                            \
  }                                                                                                \
  static const int NAME##_n = NINSTR;

// ---------------------------------------------------------------- VALU issue / dependency
TEST(fma_indep, 8,
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_fmac_f64 v[106:107], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_fmac_f64 v[110:111], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[112:113], v[102:103], v[100:101]\n v_fmac_f64 v[114:115], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[116:117], v[102:103], v[100:101]\n v_fmac_f64 v[118:119], v[102:103], v[100:101]\n")
TEST(fma_dep, 8,
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n")
TEST(fma_dep2, 8, // two chains interleaved: distance 2
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[106:107], v[102:103], v[106:107], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[106:107], v[102:103], v[106:107], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[106:107], v[102:103], v[106:107], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[106:107], v[102:103], v[106:107], v[100:101]\n")
TEST(fma_dep3, 9, // three chains: distance 3
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[106:107], v[102:103], v[106:107], v[100:101]\n"
     "v_fma_f64 v[108:109], v[102:103], v[108:109], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[106:107], v[102:103], v[106:107], v[100:101]\n"
     "v_fma_f64 v[108:109], v[102:103], v[108:109], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[104:105], v[100:101]\n v_fma_f64 v[106:107], v[102:103], v[106:107], v[100:101]\n"
     "v_fma_f64 v[108:109], v[102:103], v[108:109], v[100:101]\n")
TEST(mov32_indep, 8,
     "v_mov_b32 v104, v100\n v_mov_b32 v105, v100\n v_mov_b32 v106, v100\n v_mov_b32 v107, v100\n"
     "v_mov_b32 v108, v100\n v_mov_b32 v109, v100\n v_mov_b32 v110, v100\n v_mov_b32 v111, v100\n")
TEST(dpp_indep, 8,
     "v_mov_b32_dpp v104, v100 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v105, v101 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v106, v100 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v107, v101 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v108, v100 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v109, v101 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v110, v100 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v111, v101 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
TEST(rowdpp_indep, 8,
     "v_mov_b32_dpp v104, v100 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v105, v101 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v106, v100 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v107, v101 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v108, v100 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v109, v101 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v110, v100 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v111, v101 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
// the recurrence of the sweep: nv -> DPP (lo, hi) -> fma(bU, U, t)=nv  (s_nop 1: the VALU-write -> DPP-read hazard)
TEST(chain_dpp_fma, 4,
     "s_nop 1\n v_mov_b32_dpp v106, v104 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v107, v105 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_fma_f64 v[104:105], v[102:103], v[106:107], v[100:101]\n")
// the same with the bL fma in front: nv -> fma(bL, nv, t) -> fma(bU, U, .)
TEST(chain_dpp_fma2, 5,
     "s_nop 1\n v_mov_b32_dpp v106, v104 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_mov_b32_dpp v107, v105 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "v_fma_f64 v[108:109], v[102:103], v[104:105], v[100:101]\n"
     "v_fma_f64 v[104:105], v[102:103], v[106:107], v[108:109]\n")
TEST(add_max_dep, 8,
     "v_add_f64 v[106:107], v[104:105], -v[100:101]\n v_max_f64 v[108:109], v[108:109], |v[106:107]|\n"
     "v_add_f64 v[106:107], v[104:105], -v[100:101]\n v_max_f64 v[108:109], v[108:109], |v[106:107]|\n"
     "v_add_f64 v[106:107], v[104:105], -v[100:101]\n v_max_f64 v[108:109], v[108:109], |v[106:107]|\n"
     "v_add_f64 v[106:107], v[104:105], -v[100:101]\n v_max_f64 v[108:109], v[108:109], |v[106:107]|\n")
TEST(acc_write, 8,
     "v_accvgpr_write_b32 a0, v100\n v_accvgpr_write_b32 a1, v101\n v_accvgpr_write_b32 a2, v100\n v_accvgpr_write_b32 a3, v101\n"
     "v_accvgpr_write_b32 a4, v100\n v_accvgpr_write_b32 a5, v101\n v_accvgpr_write_b32 a6, v100\n v_accvgpr_write_b32 a7, v101\n")
TEST(mov64_indep, 8,
     "v_mov_b64 v[104:105], v[100:101]\n v_mov_b64 v[106:107], v[100:101]\n v_mov_b64 v[108:109], v[100:101]\n v_mov_b64 v[110:111], v[100:101]\n"
     "v_mov_b64 v[112:113], v[100:101]\n v_mov_b64 v[114:115], v[100:101]\n v_mov_b64 v[116:117], v[100:101]\n v_mov_b64 v[118:119], v[100:101]\n")
TEST(salu_bfm, 8,
     "s_bfm_b64 s[26:27], 5, 0\n s_bfm_b64 s[28:29], 6, 0\n s_bfm_b64 s[26:27], 7, 0\n s_bfm_b64 s[28:29], 8, 0\n"
     "s_bfm_b64 s[26:27], 9, 0\n s_bfm_b64 s[28:29], 10, 0\n s_bfm_b64 s[26:27], 11, 0\n s_bfm_b64 s[28:29], 12, 0\n")
TEST(valu_salu_mix, 8, // do SALU instructions hide behind fp64 VALU?
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n s_bfm_b64 s[26:27], 5, 0\n"
     "v_fmac_f64 v[106:107], v[102:103], v[100:101]\n s_bfm_b64 s[28:29], 6, 0\n"
     "v_fmac_f64 v[108:109], v[102:103], v[100:101]\n s_bfm_b64 s[26:27], 7, 0\n"
     "v_fmac_f64 v[110:111], v[102:103], v[100:101]\n s_bfm_b64 s[28:29], 8, 0\n")
TEST(fma_mov32_mix, 8, // does a 32-bit VALU hide behind fp64 VALU?
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_mov_b32 v120, v100\n"
     "v_fmac_f64 v[106:107], v[102:103], v[100:101]\n v_mov_b32 v121, v100\n"
     "v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_mov_b32 v122, v100\n"
     "v_fmac_f64 v[110:111], v[102:103], v[100:101]\n v_mov_b32 v123, v100\n")
TEST(nop8, 8, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
// ---------------------------------------------------------------- LDS issue (results never waited for inside the body)
TEST(ds_b64, 4, "ds_read_b64 v[104:105], v140\n ds_read_b64 v[106:107], v140 offset:512\n ds_read_b64 v[108:109], v140 offset:1024\n ds_read_b64 v[110:111], v140 offset:1536\n")
TEST(ds_2b64, 4, "ds_read2_b64 v[104:107], v140 offset1:64\n ds_read2_b64 v[108:111], v140 offset0:128 offset1:192\n"
                 "ds_read2_b64 v[112:115], v140 offset1:64\n ds_read2_b64 v[116:119], v140 offset0:128 offset1:192\n")
TEST(ds_b128, 4, "ds_read_b128 v[104:107], v141\n ds_read_b128 v[108:111], v141 offset:1024\n ds_read_b128 v[112:115], v141 offset:2048\n ds_read_b128 v[116:119], v141 offset:3072\n")
TEST(ds_b128_same, 4, // every lane the same address (a class table hit by one class)
     "ds_read_b128 v[104:107], v143\n ds_read_b128 v[108:111], v143 offset:16\n ds_read_b128 v[112:115], v143 offset:32\n ds_read_b128 v[116:119], v143 offset:48\n")
TEST(ds_2b64_same, 4,
     "ds_read2_b64 v[104:107], v143 offset1:32\n ds_read2_b64 v[108:111], v143 offset0:64 offset1:96\n"
     "ds_read2_b64 v[112:115], v143 offset1:32\n ds_read2_b64 v[116:119], v143 offset0:64 offset1:96\n")
TEST(ds_w64, 4, "ds_write_b64 v140, v[100:101]\n ds_write_b64 v140, v[100:101] offset:512\n ds_write_b64 v140, v[100:101] offset:1024\n ds_write_b64 v140, v[100:101] offset:1536\n")
TEST(ds_w128, 4, "ds_write_b128 v141, v[100:103]\n ds_write_b128 v141, v[100:103] offset:1024\n ds_write_b128 v141, v[100:103] offset:2048\n ds_write_b128 v141, v[100:103] offset:3072\n")
TEST(ds_addf64, 4, "ds_add_f64 v140, v[102:103]\n ds_add_f64 v140, v[102:103] offset:512\n ds_add_f64 v140, v[102:103] offset:1024\n ds_add_f64 v140, v[102:103] offset:1536\n")
TEST(ds_addf64_rtn, 4, "ds_add_rtn_f64 v[104:105], v140, v[102:103]\n ds_add_rtn_f64 v[106:107], v140, v[102:103] offset:512\n ds_add_rtn_f64 v[108:109], v140, v[102:103] offset:1024\n ds_add_rtn_f64 v[110:111], v140, v[102:103] offset:1536\n")
// ---------------------------------------------------------------- VMEM issue: 512-byte loads that hit L1 (v140 = lane*8), base s[26:27]
// (the salu_bfm / valu_salu_mix tests above clobber s[26:27]: these come first in the table)
TEST(gld_only, 4, "global_load_dwordx2 v[104:105], v140, s[26:27]\n global_load_dwordx2 v[106:107], v140, s[26:27] offset:512\n"
                  "global_load_dwordx2 v[108:109], v140, s[26:27] offset:1024\n global_load_dwordx2 v[110:111], v140, s[26:27] offset:1536\n")
TEST(gld_1_fma12, 13, "global_load_dwordx2 v[120:121], v140, s[26:27]\n"
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_fmac_f64 v[106:107], v[102:103], v[100:101]\n v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_fmac_f64 v[110:111], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[112:113], v[102:103], v[100:101]\n v_fmac_f64 v[114:115], v[102:103], v[100:101]\n v_fmac_f64 v[116:117], v[102:103], v[100:101]\n v_fmac_f64 v[118:119], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_fmac_f64 v[106:107], v[102:103], v[100:101]\n v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_fmac_f64 v[110:111], v[102:103], v[100:101]\n")
TEST(fma12, 12,
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_fmac_f64 v[106:107], v[102:103], v[100:101]\n v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_fmac_f64 v[110:111], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[112:113], v[102:103], v[100:101]\n v_fmac_f64 v[114:115], v[102:103], v[100:101]\n v_fmac_f64 v[116:117], v[102:103], v[100:101]\n v_fmac_f64 v[118:119], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_fmac_f64 v[106:107], v[102:103], v[100:101]\n v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_fmac_f64 v[110:111], v[102:103], v[100:101]\n")
// the same load every 52 instructions with a wait three loads back, as in k_sweep_roll (one class word per four steps)
TEST(gld_1_fma12_wait, 14, "global_load_dwordx2 v[120:121], v140, s[26:27]\n s_waitcnt vmcnt(3)\n"
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_fmac_f64 v[106:107], v[102:103], v[100:101]\n v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_fmac_f64 v[110:111], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[112:113], v[102:103], v[100:101]\n v_fmac_f64 v[114:115], v[102:103], v[100:101]\n v_fmac_f64 v[116:117], v[102:103], v[100:101]\n v_fmac_f64 v[118:119], v[102:103], v[100:101]\n"
     "v_fmac_f64 v[104:105], v[102:103], v[100:101]\n v_fmac_f64 v[106:107], v[102:103], v[100:101]\n v_fmac_f64 v[108:109], v[102:103], v[100:101]\n v_fmac_f64 v[110:111], v[102:103], v[100:101]\n")
// ---------------------------------------------------------------- the sweep step
// current order (k_sweep_reg round 1): sdwa add, 2 ds_read2_b64 + 2 ds_read_b64 (used two steps later: here the
// previous body's), dpp D, fmac bD, fmac bR, dpp U, fmac bL, fmac bU, add, max.
// registers: nv v[104:105]; t/A v[106:107]; coefficients v[110:117] (bU bD | bL bR); D v[118:119]; U v[120:121];
// R v[100:101]; old v[122:123]; dmax v[124:125]; LDS dst v[130:..] (never consumed: issue cost only)
#define STEP_LOADS_2B64 \
  "v_add_u32_sdwa v144, v143, v141 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n" \
  "ds_read2_b64 v[130:133], v143 offset1:32\n ds_read2_b64 v[134:137], v143 offset0:64 offset1:96\n" \
  "ds_read_b64 v[138:139], v140 offset:8192\n ds_read_b64 v[146:147], v140 offset:16384\n"
#define STEP_LOADS_B128 \
  "v_add_u32_sdwa v144, v143, v141 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n" \
  "ds_read_b128 v[130:133], v143\n ds_read_b128 v[134:137], v143 offset:16\n" \
  "ds_read_b64 v[138:139], v140 offset:8192\n ds_read_b64 v[146:147], v140 offset:16384\n"
#define STEP_MATH_R1 \
  "v_mov_b32_dpp v118, v100 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v119, v101 wave_shl:1 row_mask:0xf bank_mask:0xf\n" \
  "v_fmac_f64 v[106:107], v[112:113], v[118:119]\n v_fmac_f64 v[106:107], v[116:117], v[100:101]\n" \
  "v_mov_b32_dpp v120, v104 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v121, v105 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
  "v_fmac_f64 v[106:107], v[114:115], v[104:105]\n v_fma_f64 v[104:105], v[110:111], v[120:121], v[106:107]\n" \
  "v_add_f64 v[122:123], v[104:105], -v[122:123]\n v_max_f64 v[124:125], v[124:125], |v[122:123]|\n"
TEST(step_r1_2b64, 15, STEP_LOADS_2B64 "s_waitcnt lgkmcnt(8)\n" STEP_MATH_R1)
TEST(step_r1_b128, 15, STEP_LOADS_B128 "s_waitcnt lgkmcnt(8)\n" STEP_MATH_R1)
TEST(step_r1_noloads, 10, STEP_MATH_R1)
// interleaved order: finish step i (U, bL, bU, add, max) while starting step i+1 (D, bD, bR) on a second accumulator
#define STEP_MATH_IL \
  "v_mov_b32_dpp v120, v104 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v121, v105 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
  "v_fmac_f64 v[106:107], v[114:115], v[104:105]\n" \
  "v_mov_b32_dpp v118, v100 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v119, v101 wave_shl:1 row_mask:0xf bank_mask:0xf\n" \
  "v_fma_f64 v[104:105], v[110:111], v[120:121], v[106:107]\n" \
  "v_fmac_f64 v[108:109], v[112:113], v[118:119]\n" \
  "v_add_f64 v[122:123], v[104:105], -v[122:123]\n" \
  "v_fmac_f64 v[108:109], v[116:117], v[100:101]\n" \
  "v_max_f64 v[124:125], v[124:125], |v[122:123]|\n"
TEST(step_il_b128, 15, STEP_LOADS_B128 "s_waitcnt lgkmcnt(8)\n" STEP_MATH_IL)
TEST(step_il_noloads, 10, STEP_MATH_IL)
// loads spread between the arithmetic
TEST(step_il_spread, 15,
     "v_add_u32_sdwa v144, v143, v141 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
     "v_mov_b32_dpp v120, v104 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v121, v105 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
     "ds_read_b128 v[130:133], v143\n"
     "v_fmac_f64 v[106:107], v[114:115], v[104:105]\n"
     "v_mov_b32_dpp v118, v100 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v119, v101 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
     "ds_read_b128 v[134:137], v143 offset:16\n"
     "v_fma_f64 v[104:105], v[110:111], v[120:121], v[106:107]\n"
     "v_fmac_f64 v[108:109], v[112:113], v[118:119]\n"
     "ds_read_b64 v[138:139], v140 offset:8192\n"
     "v_add_f64 v[122:123], v[104:105], -v[122:123]\n"
     "v_fmac_f64 v[108:109], v[116:117], v[100:101]\n"
     "ds_read_b64 v[146:147], v140 offset:16384\n"
     "s_waitcnt lgkmcnt(8)\n"
     "v_max_f64 v[124:125], v[124:125], |v[122:123]|\n")
// window step extras on top of step_r1: s_bfm, two cndmask, second max, copy to AGPRs
TEST(step_r1_window, 21, STEP_LOADS_B128 "s_waitcnt lgkmcnt(8)\n" STEP_MATH_R1
     "s_bfm_b64 s[26:27], 7, 0\n v_accvgpr_write_b32 a0, v122\n v_accvgpr_write_b32 a1, v123\n"
     "v_cndmask_b32 v126, 0, v123, s[26:27]\n v_cndmask_b32 v127, v123, 0, s[26:27]\n"
     "v_max_f64 v[128:129], v[128:129], |v[126:127]|\n")

struct T { const char *name; void (*fn)(long long *, int, const char *); int n; };
#define E(NAME) {#NAME, NAME, NAME##_n}
int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, iters = 400;
  T tests[] = {E(gld_only), E(gld_1_fma12), E(fma12), E(gld_1_fma12_wait), E(nop8), E(fma_indep), E(fma_dep), E(fma_dep2), E(fma_dep3), E(mov32_indep), E(mov64_indep), E(dpp_indep), E(rowdpp_indep),
               E(chain_dpp_fma), E(chain_dpp_fma2), E(add_max_dep), E(acc_write), E(salu_bfm), E(valu_salu_mix), E(fma_mov32_mix),
               E(ds_b64), E(ds_2b64), E(ds_b128), E(ds_b128_same), E(ds_2b64_same), E(ds_w64), E(ds_w128), E(ds_addf64), E(ds_addf64_rtn),
               E(step_r1_2b64), E(step_r1_b128), E(step_r1_noloads), E(step_il_b128), E(step_il_noloads), E(step_il_spread), E(step_r1_window)};
  long long *d;
  hipMalloc(&d, 8 * blocks);
  char *gbuf;
  hipMalloc(&gbuf, 1 << 20);
  hipMemset(gbuf, 0, 1 << 20);
  printf("%d workgroups x 64 threads, 40 KB LDS each (four per CU), %d x 16 bodies\n", blocks, iters);
  for (const T &t : tests) {
    hipFuncSetAttribute((const void *)t.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(64), 40960, 0, d, iters, gbuf);
      hipDeviceSynchronize();
    }
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= blocks;
    const double per_body = avg / (iters * 16.0);
    printf("%-18s %2d instr/body: %7.1f cycles/body  %5.2f cycles/instr\n", t.name, t.n, per_body, per_body / t.n);
  }
  return 0;
}
