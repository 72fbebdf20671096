/* ref_intra_harness.c -- TEST INFRASTRUCTURE ONLY.  Gives the harness access to the reference's file-local refinement
 * of the intra mode search (search_intra_rdo, sort_modes: search_intra.c:47-63, :573-650) by compiling that translation unit a
 * second time, from where it lies under /root/reference, with its exported names moved out of the way -- the same device
 * as ref_me_harness.c.  Nothing of the reference is copied into this repository. */
#define kvz_luma_mode_bits          refintra_dup_luma_mode_bits
#define kvz_chroma_mode_bits        refintra_dup_chroma_mode_bits
#define kvz_search_intra_chroma_rdo refintra_dup_search_intra_chroma_rdo
#define kvz_search_cu_intra_chroma  refintra_dup_search_cu_intra_chroma
#define kvz_search_cu_intra         refintra_dup_search_cu_intra
#include "search_intra.c"
#undef kvz_luma_mode_bits
#undef kvz_chroma_mode_bits
#undef kvz_search_intra_chroma_rdo
#undef kvz_search_cu_intra_chroma
#undef kvz_search_cu_intra

/* the part of kvz_search_cu_intra that follows the rough search when rd >= 2 (search_intra.c:857-878) */
int8_t refintra_refine(encoder_state_t *state, int x_px, int y_px, int depth, kvz_pixel *orig, int32_t origstride, int8_t *intra_preds,
                       int modes_to_check, int8_t number_of_modes, int8_t modes[35], double costs[35], lcu_t *lcu)
{
  sort_modes(modes, costs, number_of_modes);
  return search_intra_rdo(state, x_px, y_px, depth, orig, origstride, intra_preds, modes_to_check, modes, costs, lcu);
}
