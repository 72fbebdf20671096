/* ref_me_harness.c -- TEST INFRASTRUCTURE ONLY.  Gives the tests access to the reference's
 * file-local motion search (hexagon_search, search_frac, calc_mvd_cost, select_mv_cand:
 * search_inter.c:195-1128) by compiling that translation unit a second time, from where it lies
 * under /root/reference, with its exported names moved out of the way.  Nothing of the reference is
 * copied into this repository; oracle/Makefile builds this file into oracle/_ref/libkvzref.so. */
#define kvz_tz_pattern_search  refme_dup_tz_pattern_search
#define kvz_tz_raster_search   refme_dup_tz_raster_search
#define kvz_cu_cost_inter_rd2  refme_dup_cu_cost_inter_rd2
#define kvz_search_cu_inter    refme_dup_search_cu_inter
#define kvz_search_cu_smp      refme_dup_search_cu_smp
#include "search_inter.c"
#undef kvz_tz_pattern_search
#undef kvz_tz_raster_search
#undef kvz_cu_cost_inter_rd2
#undef kvz_search_cu_inter
#undef kvz_search_cu_smp

#include <string.h>

/* same layouts as orc_me_pu / orc_me_params / orc_me_result (oracle/kvz_oracle.h) */
typedef struct { int16_t mv[2]; uint8_t usable; uint8_t same_ref; } me_merge_t;
typedef struct {
  int32_t x, y, width, height;
  int16_t mv_cand[2][2];
  int16_t extra_mv[2];
  int16_t num_merge_cand, reserved;
  me_merge_t merge[5];
  int16_t pad;
} me_pu_t;
typedef struct {
  int32_t lambda_cost, early_termination;
  uint32_t max_steps;
  int32_t fme_level, wpp_owf, ref_delay_px, max_ref_lcu_down, max_ref_lcu_right;
  int32_t algorithm, search_range, size_classes, mv_constraint;
  int32_t tile_x, tile_y, tile_w, tile_h;
  int32_t mv_rdo, ref_idx, refs_before, reserved;
  const struct me_cabac_s *cabac;
  const uint32_t *cost_to_beat;               /* the caller points it at THIS PU's entry (or NULL) */
} me_params_t;
typedef struct me_cabac_s { uint16_t range; uint8_t ctx[8]; uint8_t pad[6]; } me_cabac_t;
typedef struct { int32_t mv[2]; uint32_t cost, bitcost; int32_t merged, merge_idx, mv_cand, reserved; } me_result_t;

/* the hexbs path of search_pu_inter_ref (search_inter.c:1134-1300) on a fabricated encoder state:
 * one reference picture (ref_idx 0 = L0[0]), mv_rdo off.  With a tile (tile_w x tile_h at tile_x, tile_y; 0 x 0 = the
 * whole frame) the state is set up the way encoderstate.c does for a tile: state->tile->offset_x/_y, a tile frame of
 * the tile's size whose source is a sub-image of the picture, info->origin relative to the tile; references stay whole
 * frames (kvz_image_calc_sad / kvz_get_extended_block add the tile offset themselves). */
void ref_me_search_pu(const kvz_pixel *pic_y, const kvz_pixel *ref_y, int frame_w, int frame_h,
                      const me_pu_t *pu, const me_params_t *prm, me_result_t *res)
{
  static encoder_control_t ctrl;
  static encoder_state_t state;
  static encoder_state_config_frame_t frame;
  static encoder_state_config_tile_t tile;
  static videoframe_t vframe;
  memset(&ctrl, 0, sizeof(ctrl)); memset(&state, 0, sizeof(state)); memset(&frame, 0, sizeof(frame));
  memset(&tile, 0, sizeof(tile)); memset(&vframe, 0, sizeof(vframe));
  ctrl.bitdepth = 8;
  ctrl.cfg.me_early_termination = prm->early_termination;
  ctrl.cfg.me_max_steps = prm->max_steps;
  ctrl.cfg.fme_level = prm->fme_level;
  ctrl.cfg.owf = prm->wpp_owf ? 1 : 0;
  ctrl.cfg.wpp = prm->wpp_owf ? 1 : 0;
  ctrl.cfg.sao_type = prm->ref_delay_px == SAO_DELAY_PX ? 1 : 0;
  ctrl.cfg.deblock_enable = prm->ref_delay_px == DEBLOCK_DELAY_PX ? 1 : 0;
  ctrl.cfg.mv_constraint = (enum kvz_mv_constraint)prm->mv_constraint;
  const int tiled = prm->tile_w != 0 || prm->tile_h != 0;
  const int tx = tiled ? prm->tile_x : 0, ty = tiled ? prm->tile_y : 0;
  const int tw = tiled ? prm->tile_w : frame_w, th = tiled ? prm->tile_h : frame_h;
  ctrl.cfg.mv_rdo = prm->mv_rdo ? 1 : 0;
  ctrl.max_inter_ref_lcu.down = prm->max_ref_lcu_down;
  ctrl.max_inter_ref_lcu.right = prm->max_ref_lcu_right;
  vframe.width = tw; vframe.height = th;
  tile.frame = &vframe;
  tile.offset_x = tx; tile.offset_y = ty;
  for (int i = 0; i < 16; ++i) frame.ref_LX[0][i] = (uint8_t)i;
  frame.ref_LX_size[0] = 16;
  const int search_ref = prm->mv_rdo ? prm->ref_idx : 0;           /* info->ref_idx; the merge candidates' "same_ref" is relative to it */
  /* --mv-rdo (rdo.c:908-1060): the CABAC state the cost model starts from, and the reference list it codes ref_idx against */
  static image_list_t reflist;
  static int32_t pocs[16];
  memset(&reflist, 0, sizeof(reflist));
  if (prm->mv_rdo) {
    const me_cabac_t *cb = &prm->cabac[pu->reserved];
    state.cabac.range = cb->range;
    state.cabac.bits_left = 23;
    state.cabac.ctx.cu_merge_flag_ext_model.uc_state = cb->ctx[0];
    state.cabac.ctx.cu_merge_idx_ext_model.uc_state = cb->ctx[1];
    state.cabac.ctx.cu_ref_pic_model[0].uc_state = cb->ctx[2];
    state.cabac.ctx.cu_ref_pic_model[1].uc_state = cb->ctx[3];
    state.cabac.ctx.cu_mvd_model[0].uc_state = cb->ctx[4];
    state.cabac.ctx.cu_mvd_model[1].uc_state = cb->ctx[5];
    state.cabac.ctx.mvp_idx_model[0].uc_state = cb->ctx[6];
    frame.poc = 100;
    reflist.used_size = prm->refs_before > 0 ? prm->refs_before : 1;
    for (int i = 0; i < 16; ++i) pocs[i] = 99 - i;
    reflist.pocs = pocs;
    frame.ref = &reflist;
  }
  state.encoder_control = &ctrl;
  state.tile = &tile;
  state.frame = &frame;
  /* calc_mvd_cost multiplies by (int32_t)(lambda_sqrt + 0.5) */
  state.lambda_sqrt = (double)prm->lambda_cost;

  kvz_picture pic, ref;
  memset(&pic, 0, sizeof(pic)); memset(&ref, 0, sizeof(ref));
  pic.y = (kvz_pixel *)pic_y + (size_t)ty * frame_w + tx; pic.width = tw; pic.height = th; pic.stride = frame_w;   /* kvz_image_make_subimage */
  ref.y = (kvz_pixel *)ref_y; ref.width = frame_w; ref.height = frame_h; ref.stride = frame_w;

  inter_search_info_t info = {
    .state = &state, .pic = &pic, .ref = &ref, .ref_idx = 0,
    .origin = { pu->x - tx, pu->y - ty }, .width = pu->width, .height = pu->height,
    .mvd_cost_func = prm->mv_rdo ? kvz_calc_mvd_cost_cabac : calc_mvd_cost,
  };
  info.ref_idx = search_ref;
  memcpy(info.mv_cand, pu->mv_cand, sizeof(info.mv_cand));
  info.num_merge_cand = pu->num_merge_cand;
  for (int i = 0; i < pu->num_merge_cand; ++i) {
    memset(&info.merge_cand[i], 0, sizeof(info.merge_cand[i]));
    info.merge_cand[i].dir = pu->merge[i].usable ? 1 : 3;
    info.merge_cand[i].ref[0] = pu->merge[i].same_ref ? search_ref : (search_ref + 1) % 4;
    info.merge_cand[i].mv[0][0] = pu->merge[i].mv[0];
    info.merge_cand[i].mv[0][1] = pu->merge[i].mv[1];
  }
  const vector2d_t extra = { pu->extra_mv[0], pu->extra_mv[1] };

  info.best_cost = UINT32_MAX;
  if (prm->algorithm == 1) diamond_search(&info, extra, prm->max_steps);
  else if (prm->algorithm == 2) tz_search(&info, extra);
  else if (prm->algorithm == 3) search_mv_full(&info, prm->search_range, extra);
  else hexagon_search(&info, extra, prm->max_steps);
  const double inter_cost = prm->cost_to_beat ? (double)*prm->cost_to_beat : (double)MAX_INT;      /* *inter_cost, search_inter.c:1239, :1456 */
  if (prm->fme_level > 0 && info.best_cost < inter_cost) {
    search_frac(&info);
  } else if (info.best_cost < UINT32_MAX) {
    info.best_cost = kvz_image_calc_satd(&pic, &ref, info.origin.x, info.origin.y, tx + info.origin.x + (info.best_mv.x >> 2),
                                         ty + info.origin.y + (info.best_mv.y >> 2), pu->width, pu->height);   /* :1236-1248 */
    info.best_cost += info.best_bitcost * (int)(state.lambda_sqrt + 0.5);
  }
  memset(res, 0, sizeof(*res));
  res->mv[0] = info.best_mv.x; res->mv[1] = info.best_mv.y;
  res->cost = info.best_cost; res->bitcost = info.best_bitcost;
  int idx;
  for (idx = 0; idx < info.num_merge_cand; ++idx)
    if (info.merge_cand[idx].dir != 3 && info.merge_cand[idx].mv[0][0] == info.best_mv.x &&
        info.merge_cand[idx].mv[0][1] == info.best_mv.y && frame.ref_LX[0][info.merge_cand[idx].ref[0]] == search_ref) { res->merged = 1; break; }
  res->merge_idx = idx;
  if (!res->merged) res->mv_cand = select_mv_cand(&state, info.mv_cand, info.best_mv.x, info.best_mv.y, NULL);
}

/* luma of kvz_inter_recon_bipred (inter.c:430-477) + kvz_satd_any_size against the source, as search_pu_inter_bipred
 * scores a candidate pair (search_inter.c:1349-1362); two reference pictures of the frame's size; 4:0:0 so that the
 * fabricated pictures need no chroma planes.  pred_out (may be NULL) receives the w x h prediction. */
#include "inter.h"
unsigned ref_bipred_luma_satd(const kvz_pixel *pic_y, const kvz_pixel *ref0_y, const kvz_pixel *ref1_y, int frame_w, int frame_h,
                              int x, int y, int w, int h, const int16_t *mv0, const int16_t *mv1, kvz_pixel *pred_out)
{
  static encoder_control_t ctrl;
  static encoder_state_t state;
  static encoder_state_config_tile_t tile;
  static videoframe_t vframe;
  memset(&ctrl, 0, sizeof(ctrl)); memset(&state, 0, sizeof(state)); memset(&tile, 0, sizeof(tile)); memset(&vframe, 0, sizeof(vframe));
  ctrl.bitdepth = 8;
  ctrl.cfg.bipred = 1;
  ctrl.chroma_format = KVZ_CSP_400;
  vframe.width = frame_w; vframe.height = frame_h;
  tile.frame = &vframe;
  state.encoder_control = &ctrl;
  state.tile = &tile;
  kvz_picture r0, r1;
  memset(&r0, 0, sizeof(r0)); memset(&r1, 0, sizeof(r1));
  r0.y = (kvz_pixel *)ref0_y; r0.width = frame_w; r0.height = frame_h; r0.stride = frame_w;
  r1.y = (kvz_pixel *)ref1_y; r1.width = frame_w; r1.height = frame_h; r1.stride = frame_w;
  lcu_t *lcu = calloc(1, sizeof(lcu_t));
  int16_t mv[2][2] = { { mv0[0], mv0[1] }, { mv1[0], mv1[1] } };
  kvz_inter_recon_bipred(&state, &r0, &r1, x, y, w, h, mv, lcu);
  const kvz_pixel *rec = &lcu->rec.y[(y % LCU_WIDTH) * LCU_WIDTH + (x % LCU_WIDTH)];
  const unsigned cost = kvz_satd_any_size(w, h, rec, LCU_WIDTH, pic_y + (size_t)y * frame_w + x, frame_w);
  if (pred_out)
    for (int r = 0; r < h; ++r) memcpy(pred_out + (size_t)r * w, rec + r * LCU_WIDTH, (size_t)w);
  free(lcu);
  return cost;
}

/* The reference's file-local helpers of the bi-prediction search (search_pu_inter_bipred, search_inter.c:1304-1440), for the
 * harness code that serves that search from the GPU entry inside a live encode (oracle/ref_harness.c: gpu_serve_pu): the
 * MV bit cost, the AMVP choice and the MV constraint test are the reference's own. */
uint32_t refme_calc_mvd_cost(const encoder_state_t *state, int x, int y, int mv_shift, int16_t mv_cand[2][2], uint32_t *bitcost)
{
  return calc_mvd_cost(state, x, y, mv_shift, mv_cand, NULL, 0, 0, bitcost);
}

int refme_select_mv_cand(const encoder_state_t *state, int16_t mv_cand[2][2], int32_t mv_x, int32_t mv_y)
{
  return select_mv_cand(state, mv_cand, mv_x, mv_y, NULL);
}

int refme_fracmv_within_tile(const encoder_state_t *state, int origin_x, int origin_y, int width, int height, int mv_x, int mv_y)
{
  inter_search_info_t info = { .state = (encoder_state_t *)state, .origin = { origin_x, origin_y }, .width = width, .height = height };
  return fracmv_within_tile(&info, mv_x, mv_y);
}
