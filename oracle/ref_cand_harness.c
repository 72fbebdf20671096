/* ref_cand_harness.c -- TEST INFRASTRUCTURE ONLY.  Gives the tests access to the reference's file-local candidate helpers
 * (is_a0_cand_coded, is_b0_cand_coded, get_spatial_merge_candidates: inter.c:566-875) the way the reference's own unit
 * test does (tests/mv_cand_tests.c:20 includes src/inter.c): the translation unit is compiled a second time, from where
 * it lies under /root/reference, with its exported names moved out of the way.  Nothing of the reference is copied into
 * this repository; oracle/Makefile builds this file into oracle/_ref/libkvzref.so. */
#define kvz_inter_recon_bipred     refcand_dup_inter_recon_bipred
#define kvz_inter_recon_cu         refcand_dup_inter_recon_cu
#define kvz_inter_get_mv_cand      refcand_dup_inter_get_mv_cand
#define kvz_inter_get_mv_cand_cua  refcand_dup_inter_get_mv_cand_cua
#define kvz_inter_get_merge_cand   refcand_dup_inter_get_merge_cand
#include "inter.c"
#undef kvz_inter_recon_bipred
#undef kvz_inter_recon_cu
#undef kvz_inter_get_mv_cand
#undef kvz_inter_get_mv_cand_cua
#undef kvz_inter_get_merge_cand

int ref_is_a0_cand_coded(int x, int y, int width, int height) { return is_a0_cand_coded(x, y, width, height) ? 1 : 0; }
int ref_is_b0_cand_coded(int x, int y, int width, int height) { return is_b0_cand_coded(x, y, width, height) ? 1 : 0; }

/* tests/mv_cand_tests.c:26-49: an LCU whose CUs are all inter; out[0..4] = the indices in lcu.cu of b0 b1 b2 a0 a1 (-1: none) */
void ref_spatial_merge_candidate_indices(int x, int y, int width, int height, int pic_w, int pic_h, int *out)
{
  static lcu_t lcu;
  memset(&lcu, 0, sizeof(lcu));
  for (size_t i = 0; i < sizeof(lcu.cu) / sizeof(cu_info_t); i++) lcu.cu[i].type = CU_INTER;
  merge_candidates_t cand = { { 0, 0 }, { 0, 0, 0 }, 0, 0 };
  get_spatial_merge_candidates(x, y, width, height, pic_w, pic_h, &lcu, &cand);
  const cu_info_t *p[5] = { cand.b[0], cand.b[1], cand.b[2], cand.a[0], cand.a[1] };
  for (int i = 0; i < 5; ++i) out[i] = p[i] ? (int)(p[i] - lcu.cu) : -1;
}
