/*
 * kvz_oracle.h -- CPU restatement of Kvazaar's `generic` block kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kvazaar_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/src) whose semantics it restates.  The arithmetic is
 * re-derived (e.g. the DCT is a plain integer matrix product, the Hadamard
 * transforms are H*D*H^T), not transcribed; bit-exactness against the
 * compiled reference is pinned by tests/test_oracle_vs_ref.py (in the build
 * container, via oracle/_ref) and by the golden fixtures under tests/golden/.
 *
 * PARITY PINNED: by (1) the reference's own known-answer values
 * (tests/satd_tests.c:109,127,146; tests/sad_tests.c:121-259;
 * tests/coeff_sum_tests.c:29-43) and (2) outputs of the reference's generic
 * strategy compiled from /root/reference by oracle/Makefile (oracle/_ref).
 */
#ifndef KVZ_ORACLE_H_
#define KVZ_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t orc_pixel;   /* kvz_pixel, KVZ_BIT_DEPTH == 8 (kvazaar.h:72-77) */
typedef int16_t orc_coeff;   /* coeff_t (global.h:99) */

#define ORC_LCU_WIDTH 64
#define ORC_EXT_BLOCK_W_LUMA 71   /* search_inter.h:46 */
#define ORC_IPOL_ROWS (ORC_EXT_BLOCK_W_LUMA + 1)

/* ---- picture group (strategies/generic/picture-generic.c) ---- */
orc_pixel orc_fast_clip_16bit_to_pixel(int16_t value);             /* :30-48 */
orc_pixel orc_fast_clip_32bit_to_pixel(int32_t value);             /* :52-70 */
unsigned orc_reg_sad(const orc_pixel *d1, const orc_pixel *d2, int w, int h,
                     unsigned stride1, unsigned stride2);          /* :86-99 */
unsigned orc_sad_nxn(int n, const orc_pixel *b1, const orc_pixel *b2);   /* :460-486 */
unsigned orc_satd_4x4(const orc_pixel *b1, const orc_pixel *b2);         /* :189-196 */
unsigned orc_satd_4x4_subblock(const orc_pixel *b1, int s1,
                               const orc_pixel *b2, int s2);       /* :201-213 */
unsigned orc_satd_8x8_subblock(const orc_pixel *b1, int s1,
                               const orc_pixel *b2, int s2);       /* :240-328 */
unsigned orc_satd_nxn(int n, const orc_pixel *b1, const orc_pixel *b2);  /* strategies-picture.h:40-56 */
unsigned orc_satd_any_size(int w, int h, const orc_pixel *b1, int s1,
                           const orc_pixel *b2, int s2);           /* strategies-picture.h:62-100 */
/* preds[k] lives at preds + k*pred_stride (1024 in the reference: pred_buffer) */
void orc_sad_nxn_dual(int n, const orc_pixel *preds, size_t pred_stride,
                      const orc_pixel *orig, unsigned costs[2]);   /* :497-519 */
void orc_satd_nxn_dual(int n, const orc_pixel *preds, size_t pred_stride,
                       const orc_pixel *orig, unsigned costs[2]);  /* :357-390 */
void orc_satd_any_size_quad(int w, int h, const orc_pixel *const preds[4], int stride,
                            const orc_pixel *orig, int orig_stride,
                            unsigned costs[4]);                    /* :392-456 */
unsigned orc_pixels_calc_ssd(const orc_pixel *ref, const orc_pixel *rec,
                             int ref_stride, int rec_stride, int width); /* :521-536 */
/* bipred blend, flattened: one plane of w x h; s0/s1 are either 14-bit
 * samples (hi_prec != 0) or pixels (hi_prec == 0, shifted << 6).  :538-588 */
void orc_bipred_blend_plane(int w, int h,
                            int hi_prec0, const int16_t *hp0, const orc_pixel *px0, int stride0,
                            int hi_prec1, const int16_t *hp1, const orc_pixel *px1, int stride1,
                            orc_pixel *dst, int dst_stride);

/* kvz_image_calc_sad / kvz_image_calc_satd (image.c:455-486, :488-545):
 * the reference block may lie (partly) outside the frame; outside pixels are
 * edge replicated (image.c:320-444 / ipol-generic.c:731-784).               */
unsigned orc_image_calc_sad(const orc_pixel *pic, int pic_stride,
                            const orc_pixel *ref, int ref_stride, int ref_w, int ref_h,
                            int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh);
unsigned orc_image_calc_satd(const orc_pixel *pic, int pic_stride,
                             const orc_pixel *ref, int ref_stride, int ref_w, int ref_h,
                             int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh);

/* batched ME costs of one CTU (check_mv_cost, search_inter.c:195-232, over image.c:455-486) */
void orc_ctu_sad_grid(const orc_pixel *pic, int pic_stride, int pic_w, int pic_h,
                      const orc_pixel *ref, int ref_stride, int ref_w, int ref_h,
                      int ctu_x, int ctu_y, int mvx, int mvy,
                      const int16_t *mv_offsets, int n_mv, uint32_t *costs /* [n_mv][85] */);

/* ---- dct group (strategies/generic/dct-generic.c) ---- */
enum orc_tr_kind { ORC_DCT = 0, ORC_IDCT = 1, ORC_DST = 2, ORC_IDST = 3 };
/* n in {4,8,16,32}; DST only for n==4.  :567-617 */
void orc_transform(int kind, int n, const int16_t *in, int16_t *out);
/* the integer transform matrix row-major [n][n] (dct-generic.c:26-108) */
const int16_t *orc_dct_matrix(int n);
const int16_t *orc_dst4_matrix(void);

/* ---- quant group (strategies/generic/quant-generic.c) ---- */
typedef struct {
  int32_t qp;              /* state->qp */
  int32_t slice_is_intra;  /* state->frame->slicetype == KVZ_SLICE_I */
  int32_t signhide;        /* encoder->cfg.signhide_enable */
  int32_t scaling_list;    /* encoder->scaling_list.enable (0 = flat) */
  const int32_t *quant_coeff;    /* [w*h] when scaling_list, else NULL */
  const int32_t *dequant_coeff;  /* [w*h] when scaling_list, else NULL */
} orc_quant_params;
int32_t orc_get_scaled_qp(int type, int qp, int qp_offset);        /* transform.c:129-143 */
const uint32_t *orc_scan_order(int scan_idx, int log2_size);       /* tables.c kvz_g_sig_last_scan */
void orc_quant(const orc_quant_params *p, const orc_coeff *coef, orc_coeff *q_coef,
               int w, int h, int type, int scan_idx, int block_is_intra);  /* :37-163 */
void orc_dequant(const orc_quant_params *p, const orc_coeff *q_coef, orc_coeff *coef,
                 int w, int h, int type, int block_is_intra);       /* :279-321 */
uint32_t orc_coeff_abs_sum(const orc_coeff *c, size_t len);         /* :323-330 */
/* rdoq disabled path of kvz_quantize_residual_generic (:180-273).
 * color: 0 Y, 1 U, 2 V.  Returns has_coeffs. */
int orc_quantize_residual(const orc_quant_params *p, int cu_is_intra, int width, int color,
                          int scan_order, int use_trskip, int in_stride, int out_stride,
                          const orc_pixel *ref_in, const orc_pixel *pred_in,
                          orc_pixel *rec_out, orc_coeff *coeff_out);

/* ---- ipol group (strategies/generic/ipol-generic.c) ---- */
extern const int8_t orc_luma_filter[4][8];    /* filter.c:54-60 */
extern const int8_t orc_chroma_filter[8][4];  /* filter.c:62-72 */
void orc_sample_quarterpel_luma(const orc_pixel *src, int src_stride, int w, int h,
                                orc_pixel *dst, int dst_stride, const int16_t mv[2]);   /* :122-157 */
void orc_sample_14bit_quarterpel_luma(const orc_pixel *src, int src_stride, int w, int h,
                                      int16_t *dst, int dst_stride, const int16_t mv[2]); /* :159-190 */
void orc_sample_octpel_chroma(const orc_pixel *src, int src_stride, int w, int h,
                              orc_pixel *dst, int dst_stride, const int16_t mv[2]);     /* :660-695 */
void orc_sample_14bit_octpel_chroma(const orc_pixel *src, int src_stride, int w, int h,
                                    int16_t *dst, int dst_stride, const int16_t mv[2]); /* :697-728 */

typedef struct {
  int16_t hor[5][ORC_IPOL_ROWS * ORC_LCU_WIDTH];   /* hor_intermediate */
  int16_t cols[5][ORC_IPOL_ROWS];                   /* hor_first_cols */
} orc_ipol_state;
/* step 0..3 = hpel hor/ver, hpel diag, qpel hor/ver, qpel diag (:192-658).
 * filtered = 4 blocks of 64*64 pixels (stride 64). */
void orc_filter_frac_blocks(int step, const orc_pixel *src, int src_stride, int w, int h,
                            orc_pixel *filtered /*[4][64*64]*/, orc_ipol_state *st,
                            int fme_level, int hpel_off_x, int hpel_off_y);
/* kvz_get_extended_block_generic (:731-784) into a caller buffer of
 * (w+filter_size)*(h+filter_size) bytes; returns 1 if the copy was needed
 * (malloc_used in the reference), 0 if the window was inside the frame (then
 * nothing is written).  *inside_off receives the offset of `buffer` in ref. */
int orc_get_extended_block(int xpos, int ypos, int mv_x, int mv_y, int off_x, int off_y,
                           const orc_pixel *ref, int ref_w, int ref_h, int filter_size,
                           int w, int h, orc_pixel *out, long *inside_off);

/* ---- caller-level composite used by the fused GPU kernel's parity test:
 * search_frac's (search_inter.c:965-1128) filter + quad-SATD sequence without
 * the MV bit costs: returns the 1+16 SATD costs it evaluates for the block at
 * (x,y) size w x h with integer mv (mvx,mvy) (full-pel units), choosing the
 * best hpel offset by SATD alone.  costs_out[0] = integer position,
 * [1..8] hpel, [9..16] qpel.  best_out = {best hpel index 0..8, best qpel 0..8}. */
void orc_search_frac_costs(const orc_pixel *pic, int pic_stride,
                           const orc_pixel *ref, int ref_w, int ref_h,
                           int x, int y, int w, int h, int mvx, int mvy,
                           unsigned costs_out[17], int best_out[2]);

/* ---- intra group: src/strategies/generic/intra-generic.c + src/intra.c ----
 * SURVEY.md section 8(f) row 2.  A PU's reference pixels are the reference's
 * kvz_intra_ref (intra.h:35-38): left[0] == top[0] == the top-left corner,
 * left[1..2N] / top[1..2N] the neighbours. */
typedef struct { orc_pixel left[2 * 32 + 1]; orc_pixel top[2 * 32 + 1]; } orc_intra_ref;
/* kvz_angular_pred_generic (intra-generic.c:37-145), mode 2..34 */
void orc_angular_pred(int log2_width, int mode, const orc_pixel *ref_above, const orc_pixel *ref_left, orc_pixel *dst);
/* kvz_intra_pred_planar_generic (intra-generic.c:155-189) */
void orc_intra_pred_planar(int log2_width, const orc_pixel *ref_top, const orc_pixel *ref_left, orc_pixel *dst);
/* intra_filter_reference (intra.c:164-192) */
void orc_intra_filter_reference(int log2_width, const orc_intra_ref *ref, orc_intra_ref *filtered);
/* kvz_intra_predict (intra.c:281-331): reference smoothing decision, planar, DC (+ edge
 * filter), angular (+ boundary post-process of modes 10 / 26).  is_luma = (color == COLOR_Y). */
void orc_intra_predict(const orc_intra_ref *ref, int log2_width, int mode, int is_luma, int filter_boundary, orc_pixel *dst);
/* the costs search_intra_rough (search_intra.c:404-520) can ask for: SATD (satd_NxN) and SAD
 * (sad_NxN) of every mode 0..34 against the N x N original block (contiguous). */
void orc_intra_rough_costs(const orc_intra_ref *ref, int log2_width, int filter_boundary, const orc_pixel *orig,
                           unsigned satd_out[35], unsigned sad_out[35]);

/* kvz_intra_build_reference (intra.c:334-588) for the PU of `color` (0 Y, 1 U, 2 V) at luma position
 * (luma_x, luma_y) of a pic_w x pic_h (luma) picture, read from the not yet deblocked reconstruction plane of that
 * colour (stride in pixels of the plane).  Entries past 2N of both arrays are left alone (_many zeroes them). */
void orc_intra_build_reference(int log2_width, int color, const orc_pixel *rec, int stride, int pic_w, int pic_h,
                               int luma_x, int luma_y, orc_intra_ref *out);
void orc_intra_build_reference_many(int log2_width, int color, const orc_pixel *rec, int stride, int pic_w, int pic_h,
                                    const int32_t *xy, size_t count, orc_intra_ref *out);

/* ---- integer + fractional motion search of one PU against one reference picture:
 * hexagon_search (search_inter.c:690-778: select_starting_point :282-307, early_terminate
 * :415-460, check_mv_cost :195-232) followed by search_frac (:965-1128) with the MV bit
 * costs of calc_mvd_cost (:373-412), as search_pu_inter_ref (:1134-1300) runs them for
 * --me hexbs.  SURVEY.md section 8(f) row 1.  The encoder state these functions read is
 * flattened into the two structs below (no tiles: tile offset 0; mv_rdo off). */
typedef struct {
  int16_t mv[2];        /* merge_cand[i].mv[dir - 1], quarter-pel */
  uint8_t usable;       /* merge_cand[i].dir != 3 */
  uint8_t same_ref;     /* state->frame->ref_LX[dir - 1][merge_cand[i].ref[dir - 1]] == ref_idx */
} orc_me_merge;
typedef struct {
  int32_t x, y, width, height;   /* PU inside the picture; width, height multiples of 8, 8..64 */
  int16_t mv_cand[2][2];         /* AMVP candidates (kvz_inter_get_mv_cand), quarter-pel */
  int16_t extra_mv[2];           /* start vector taken from the co-located CU (:1190-1206), quarter-pel */
  int16_t num_merge_cand;        /* 0..5 */
  int16_t reserved;              /* mv_rdo: index of the PU's CABAC snapshot in orc_me_params.cabac */
  orc_me_merge merge[5];
  int16_t pad;
} orc_me_pu;                     /* 64 bytes */
typedef struct {
  int32_t lambda_cost;           /* (int32_t)(state->lambda_sqrt + 0.5) */
  int32_t early_termination;     /* cfg.me_early_termination: 0 off, 1 on, 2 sensitive */
  uint32_t max_steps;            /* cfg.me_max_steps */
  int32_t fme_level;             /* cfg.fme_level 0..4 */
  int32_t wpp_owf;               /* cfg.owf && cfg.wpp: enforce fracmv_within_tile's availability rule (:95-139) */
  int32_t ref_delay_px;          /* SAO_DELAY_PX (sao on), DEBLOCK_DELAY_PX (deblock only) or 0 */
  int32_t max_ref_lcu_down, max_ref_lcu_right;   /* ctrl->max_inter_ref_lcu */
  int32_t algorithm;             /* cfg.ime_algorithm: 0 hexbs (hexagon_search), 1 dia (diamond_search :796-883), 2 tz (tz_search :595-672),
                                    3 full (search_mv_full :886-962) */
  int32_t search_range;          /* algorithm 3: 8, 16, 32 or 64 (search_inter.c:1208-1215) */
  int32_t size_classes;          /* launch hint of the GPU entry; unused here */
  int32_t mv_constraint;         /* cfg.mv_constraint (kvazaar.h:113-119): the branches of fracmv_within_tile :142-171 */
  int32_t tile_x, tile_y;        /* state->tile->offset_x / _y in the picture */
  int32_t tile_w, tile_h;        /* state->tile->frame->width / height; 0 x 0 = the picture is one tile */
  int32_t mv_rdo;                /* cfg.mv_rdo: MV bits from the CABAC model, kvz_calc_mvd_cost_cabac (rdo.c:908-1060) */
  int32_t ref_idx;               /* info->ref_idx of the reference picture searched (coded when refs_before > 1) */
  int32_t refs_before;           /* pictures of state->frame->ref with poc < the current poc (rdo.c:990-998) */
  int32_t reserved;
  const struct orc_me_cabac *cabac;   /* mv_rdo: snapshots of state->cabac; a PU uses entry pu->reserved */
  const uint32_t *cost_to_beat;  /* NULL, or one entry per PU of a batch: *inter_cost on entry to search_pu_inter_ref (:1239) */
} orc_me_params;                 /* 96 bytes */
typedef struct orc_me_cabac {    /* what kvz_calc_mvd_cost_cabac reads of cabac_data_t (cabac.h:41-88) */
  uint16_t range;                /* .range */
  uint8_t ctx[8];                /* uc_state of cu_merge_flag_ext_model, cu_merge_idx_ext_model, cu_ref_pic_model[0], [1],
                                    cu_mvd_model[0], [1], mvp_idx_model[0], unused */
  uint8_t pad[6];
} orc_me_cabac;                  /* 16 bytes */
typedef struct {
  int32_t mv[2];                 /* info->best_mv, quarter-pel */
  uint32_t cost, bitcost;        /* info->best_cost, info->best_bitcost */
  int32_t merged, merge_idx;     /* the match loop of :1253-1266 */
  int32_t mv_cand;               /* select_mv_cand(..., NULL) when not merged (:1268-1273), else 0 */
  int32_t reserved;
} orc_me_result;
void orc_search_pu(const orc_pixel *pic, int pic_stride, const orc_pixel *ref, int ref_w, int ref_h,
                   const orc_me_pu *pu, const orc_me_params *prm, orc_me_result *res);
/* a batch: PU i is searched with prm->cost_to_beat[i] as the cost to beat (orc_search_pu alone: none) */
void orc_search_pu_many(const orc_pixel *pic, int pic_stride, const orc_pixel *ref, int ref_w, int ref_h,
                        const orc_me_pu *pus, size_t count, const orc_me_params *prm, orc_me_result *res);

/* ---- SAO group: src/strategies/generic/sao-generic.c, src/sao.c.  SURVEY.md section 8(f) row 4.
 * Blocks are contiguous (stride = block_width), as the callers in sao.c blit them. ---- */
typedef struct {                 /* sao_info_t (sao.h:42-50) */
  int32_t type;                  /* 0 none, 1 band, 2 edge */
  int32_t eo_class;
  int32_t ddistortion, merge_left_flag, merge_up_flag;
  int32_t band_position[2];
  int32_t offsets[10];
} orc_sao_info;
/* sao_edge_ddistortion_generic (sao-generic.c:46-77) */
int orc_sao_edge_ddistortion(const orc_pixel *orig, const orc_pixel *rec, int bw, int bh, int eo_class, const int offsets[5]);
/* calc_sao_edge_dir_generic (:80-109): accumulates into cat_sum_cnt */
void orc_calc_sao_edge_dir(const orc_pixel *orig, const orc_pixel *rec, int eo_class, int bw, int bh, int cat_sum_cnt[2][5]);
/* sao_reconstruct_color_generic (:112-154) incl. kvz_calc_sao_offset_array (sao.c:164-180); color 0 Y, 1 U, 2 V */
void orc_sao_reconstruct_color(const orc_pixel *rec, orc_pixel *new_rec, const orc_sao_info *sao, int stride, int new_stride,
                               int bw, int bh, int color);
/* sao_band_ddistortion_generic (:157-183) */
int orc_sao_band_ddistortion(const orc_pixel *orig, const orc_pixel *rec, int bw, int bh, int band_pos, const int sao_bands[4]);
/* calc_sao_bands (sao.c:247-261): accumulates into sao_bands */
void orc_calc_sao_bands(const orc_pixel *orig, const orc_pixel *rec, int bw, int bh, int sao_bands[2][32]);

/* ---- bi-prediction candidate cost (search_pu_inter_bipred, search_inter.c:1304-1440): the luma of
 * kvz_inter_recon_bipred (inter.c:430-477: per reference a 14-bit quarter-pel sample, inter.c:86-122, when its vector is
 * fractional, else the clamped pixels, inter.c:277-298, << 6; blended by inter_recon_bipred_generic) scored with
 * satd_any_size against the source block.  mv0 / mv1 quarter-pel; ref0 / ref1 planes of one size.  out (may be NULL)
 * receives the w x h prediction. */
unsigned orc_bipred_luma_satd(const orc_pixel *pic, int pic_stride, const orc_pixel *ref0, const orc_pixel *ref1, int ref_w, int ref_h,
                              int x, int y, int w, int h, const int16_t mv0[2], const int16_t mv1[2], orc_pixel *out);

/* ---- deblocking: src/filter.c:83-779 (kvz_filter_deblock_lcu over every LCU of a frame).  SURVEY.md section 8(f) row 4.
 * The frame is filtered the way the HEVC process is specified -- every vertical edge, then every horizontal edge --
 * which is what the reference's LCU-by-LCU order with its deferred "rightmost 4 pixels" (filter.c:711-779) computes.
 * cus: one record per 4x4 SCU, row-major, ceil(width/4) per row: the cu_info_t fields (cu.h:117-153) the filter reads. */
typedef struct {
  uint8_t type;                  /* cu_type_t: 1 intra, 2 inter */
  uint8_t depth, part_size, tr_depth;
  uint8_t cbf_y;                 /* cbf_is_set(cbf, tr_depth, COLOR_Y) */
  uint8_t mv_dir;                /* inter: 1 L0, 2 L1, 3 both */
  uint8_t qp;
  uint8_t reserved;
  int16_t mv[2][2];
  uint8_t mv_ref[2];
  uint8_t pad[2];
} orc_cu_info;
typedef struct {
  int32_t beta_offset_div2, tc_offset_div2;   /* cfg.deblock_beta, cfg.deblock_tc */
  int32_t qp;                    /* state->qp (used when per_cu_qp == 0) */
  int32_t frame_qp;              /* state->frame->QP */
  int32_t per_cu_qp;             /* encoder_control->max_qp_delta_depth >= 0 */
  int32_t slice_is_b;            /* state->frame->slicetype == KVZ_SLICE_B */
  int32_t chroma;                /* 0: 4:0:0, luma only; 1: 4:2:0 */
  int32_t reserved;
  uint8_t ref_LX[2][16];         /* state->frame->ref_LX */
} orc_deblock_params;
void orc_deblock_frame(orc_pixel *y, int stride_y, orc_pixel *u, orc_pixel *v, int stride_c, int width, int height,
                       const orc_cu_info *cus, const orc_deblock_params *prm);

/* ---- AMVP / merge candidate derivation: src/inter.c:546-1446 (kvz_inter_get_mv_cand :1209-1240,
 * kvz_inter_get_merge_cand :1314-1446) and the start vector of search_pu_inter_ref (search_inter.c:1190-1206).
 * SURVEY.md section 8(f) row 1, the "driver" half: what a host has to derive between two dependency fronts.
 * The encoder state is flattened: the current (tile) picture's CUs as one orc_cu_info per 4x4 SCU (what lcu->cu
 * holds while an LCU is searched: decided neighbours, type 0 = not set), the collocated picture's the same way. */
typedef struct {
  int32_t poc;                   /* state->frame->poc */
  int32_t slice_is_b;            /* state->frame->slicetype == KVZ_SLICE_B */
  int32_t tmvp_enable;           /* cfg.tmvp_enable */
  int32_t num_refs;              /* state->frame->ref->used_size */
  int32_t ref_pocs[16];          /* state->frame->ref->pocs */
  uint8_t ref_LX[2][16];         /* state->frame->ref_LX */
  uint8_t ref_LX_size[2];        /* state->frame->ref_LX_size */
  uint8_t pad[2];
  int32_t col_ref_pocs[16];      /* state->frame->ref->images[c]->ref_pocs, c = ref_LX[0][0]: the collocated picture */
  uint8_t col_ref_LX[2][16];     /* state->frame->ref->ref_LXs[c] */
  int32_t pic_width, pic_height; /* state->tile->frame->width / height */
  int32_t in_width, in_height;   /* encoder_control->in.width / height (bounds of the temporal candidates, inter.c:747,763) */
  int32_t tile_x, tile_y;        /* state->tile->offset_x / _y: only the start vector's lookup adds them (search_inter.c:1193-1194) */
  int32_t ref_idx;               /* info->ref_idx: which picture of state->frame->ref is searched */
  int32_t cus_stride;            /* records per row of cus */
  int32_t col_stride;            /* records per row of col_cus / ref_cus (cu_array_t: the width rounded up to whole LCUs / 4) */
  int32_t reserved;
} orc_inter_params;              /* 252 bytes */
typedef struct {                 /* inter_merge_cand_t (inter.h:36-41) */
  uint8_t dir;                   /* 1 L0, 2 L1, 3 both */
  uint8_t ref[2];                /* index in L0 / L1 */
  uint8_t pad;
  int16_t mv[2][2];
} orc_merge_cand;                /* 12 bytes */
/* kvz_inter_get_mv_cand for the PU at (x, y) of the tile picture, list `reflist`, index `lx_idx` in that list */
void orc_inter_get_mv_cand(const orc_cu_info *cus, const orc_cu_info *col_cus, const orc_inter_params *p,
                           int x, int y, int width, int height, int reflist, int lx_idx, int16_t mv_cand[2][2]);
/* kvz_inter_get_merge_cand; fields of `out` the reference leaves unwritten are zero */
int orc_inter_get_merge_cand(const orc_cu_info *cus, const orc_cu_info *col_cus, const orc_inter_params *p,
                             int x, int y, int width, int height, int use_a1, int use_b1, orc_merge_cand out[5]);
/* What search_pu_inter + search_pu_inter_ref derive before the search of picture p->ref_idx, for every PU: in = x, y (PICTURE
 * coordinates; the tile offset is subtracted here), width, height and pad (bit 0: A1 barred, bit 1: B1 barred -- the second PU of a two-PU CU, search_inter.c:1470-1475);
 * out = mv_cand, extra_mv (ref_cus = the SCU map of picture ref_idx, may be NULL), num_merge_cand, merge[] as
 * calc_mvd_cost reads them; merge_out (may be NULL) = the five inter_merge_cand_t per PU */
void orc_inter_candidates(const orc_cu_info *cus, const orc_cu_info *col_cus, const orc_cu_info *ref_cus, const orc_inter_params *p,
                          orc_me_pu *pus, size_t count, orc_merge_cand *merge_out);

#ifdef __cplusplus
}
#endif
/* the per-block functions over `count` contiguous blocks (whole-launch parity checks, one range per host thread) */
void orc_cost_nxn_many(int satd, int n, const orc_pixel *b1, const orc_pixel *b2, size_t count, unsigned *costs);
void orc_transform_many(int kind, int n, const int16_t *in, int16_t *out, size_t count);
/* the CABAC tables of ITU-T H.265 9.3.4.3 as the oracle restates them (checked against the reference's in tests): kind 0
 * rangeTabLps[state][quarter] (i = 4 state + q), 1 next state after an MPS, 2 after an LPS (index uc_state), 3 renormalisation shifts */
int orc_cabac_table(int kind, int i);
void orc_quantize_residual_many(const orc_quant_params *p, int cu_is_intra, int width, int color, int scan_order, int use_trskip,
                                const orc_pixel *ref_in, const orc_pixel *pred_in, orc_pixel *rec_out, orc_coeff *coeff_out,
                                int32_t *has_coeffs, size_t count);

/* the reference's own unit test of the candidate helpers (tests/mv_cand_tests.c:26-260), see kvz_oracle.c */
int orc_is_a0_cand_coded(int x, int y, int width, int height);
int orc_is_b0_cand_coded(int x, int y, int width, int height);
void orc_spatial_merge_candidate_indices(int x, int y, int width, int height, int pic_w, int pic_h, int *out);

#endif
