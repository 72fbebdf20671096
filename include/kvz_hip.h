/*
 * kvz_hip.h -- C ABI of libkvzhip.so: Kvazaar's per-CTU block kernels as
 * hand-written HIP kernels for MI355X (gfx950), exposed
 *
 *   (1) as a "hip" strategy for Kvazaar's strategyselector plugin API
 *       (reference: src/strategyselector.h:86-87 and the per-group
 *       registration hooks kvz_strategy_register_<group>_<isa>, e.g.
 *       src/strategies/avx2/picture-avx2.c:1224-1258), and
 *   (2) as batched entry points on device-resident buffers -- the form the
 *       kernels really implement and the one that is measured.
 *
 * All paths cited below are relative to the reference's src/ directory.
 * Plain C: pointers and sizes only.  Every extern symbol is kvz_-prefixed
 * (reference rule: tests/test_external_symbols.sh:7).
 *
 * Conventions
 *   kvz_pixel = uint8_t (KVZ_BIT_DEPTH 8, kvazaar.h:72-77), coeff_t = int16_t
 *   (global.h:99).  Batched entries take DEVICE pointers and a stream; they are
 *   asynchronous (enqueue only) and return KVZ_HIP_OK or a negative error code.
 *   Strategy (per-call) entries take HOST pointers exactly like the reference's
 *   typedefs and are complete (results visible to the host) on return.
 *   There is no CPU fallback anywhere: without a usable GPU kvz_hip_init fails,
 *   the registration hooks return 0 (=> kvz_strategyselector_init fails, as for
 *   any strategy that cannot register: strategyselector.c:54-95) and the batched
 *   entries return KVZ_HIP_ERR_NO_DEVICE.
 */
#ifndef KVZ_HIP_H_
#define KVZ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define KVZ_HIP_API __attribute__((visibility("default")))
#else
#define KVZ_HIP_API
#endif

typedef uint8_t kvz_hip_pixel;
typedef int16_t kvz_hip_coeff;
typedef void *kvz_hip_stream;        /* hipStream_t; any HIP stream of the current device is accepted.  NULL = the library's
                                      * default stream of the calling thread's current device: a BLOCKING stream, i.e. ordered
                                      * like the legacy default stream -- work enqueued with NULL starts after everything the host
                                      * queued on the legacy default stream before the call (hipMemcpy, a framework's default
                                      * stream) and is waited for by what the host queues there afterwards.  Streams made by
                                      * kvz_hip_stream_create() are non-blocking: inputs produced on another stream must be
                                      * complete (or ordered by an event, kvz_hip_stream_wait_event) before the entry is called. */

enum {
  KVZ_HIP_OK = 0,
  KVZ_HIP_ERR_NO_DEVICE = -1,       /* no gfx950 device / runtime failed to initialise */
  KVZ_HIP_ERR_INVALID = -2,         /* bad argument (size not supported, NULL pointer ...) */
  KVZ_HIP_ERR_RUNTIME = -3          /* a HIP call failed; see kvz_hip_last_error() */
};

/* ------------------------------------------------------------------ */
/* context                                                            */
/* ------------------------------------------------------------------ */
/* One context per device, all reachable from ONE process (the reference is a single process whose strategy
 * pointers are process-global, strategies/strategies-picture.c:33-64, and whose workers are pthreads,
 * threadqueue.c:263).  Every host thread has a current device, as with hipSetDevice: kvz_hip_init(device) creates
 * that device's context if needed (idempotent) and makes it the calling thread's current device; kvz_hip_set_device
 * switches the calling thread; a thread that never chose uses the process default = the first device initialised.
 * Every entry below runs on the calling thread's current device, so its pointer arguments and its stream must
 * belong to that device.  kvz_hip_init(-1): keep the thread's current device / the process default if there is one,
 * else $KVZ_HIP_DEVICE, else device 0 -- what the registration hooks call, the analogue of set_hardware_flags(),
 * strategyselector.c:452.  An index >= kvz_hip_device_count() fails with KVZ_HIP_ERR_INVALID. */
KVZ_HIP_API int kvz_hip_init(int device);
KVZ_HIP_API int kvz_hip_set_device(int device);      /* binds the calling thread; initialises the device on first use */
KVZ_HIP_API int kvz_hip_get_device(void);            /* the calling thread's current device, -1 before any kvz_hip_init */
KVZ_HIP_API void kvz_hip_shutdown(void);             /* destroys every context */
KVZ_HIP_API int kvz_hip_device_count(void);
KVZ_HIP_API const char *kvz_hip_last_error(void);    /* text of the calling thread's last failure */
KVZ_HIP_API const char *kvz_hip_device_name(void);   /* of the calling thread's current device */
/* Version of this header's ABI as the library was built (layouts of the kvz_hip_* structs, entry signatures); a host
 * compares it with the KVZ_HIP_ABI_VERSION it was compiled against before it registers the strategies. */
#define KVZ_HIP_ABI_VERSION 4   /* 2: per-device contexts, kvz_hip_me_params with tile / mv-constraint / mv-rdo fields;
                                   3: kvz_hip_me_params.cost_to_beat (96 bytes), candidate derivation and intra reference entries;
                                   4: kvz_hip_me_params.n_cabac (was reserved), the search service, kvz_hip_halo_exchange */
KVZ_HIP_API int kvz_hip_abi_version(void);

/* Launch-geometry / kernel-selection knobs for A/B runs (tools/bench_all.py --tune key=v1,v2);
 * value < 0 restores the built-in default.  Keys: "{sad,satd8,dct,dct16,idct16,dct32,idct32,qr,qr16,
 * qr32}_wgs_per_cu" (workgroups per CU of the streaming grids),
 * "qr4_lane_kernel", "qr8_reg_kernel", "qr_tile_kernel" (0: the LDS butterfly kernel instead of the register / matrix-core ones), "dct4_tile" (0: the LDS butterfly kernel for 4x4 transforms), "sao_edge_fast" (0/1),
 * "intra_rough_waves" (4/8 waves per workgroup of the rough search), "pair_wave_kernel" (0/1: one wave per
 * descriptor for frame-level pair batches of up to 4096 descriptors), "wg_chunk_min_wgs" (the workgroup-per-descriptor
 * kernels of the sampling / fractional search entries take up to 64 descriptors per workgroup once the list would give
 * more workgroups than this; 0: always one), "pair_satd_threads" (128 / 256 / 512 threads per workgroup of the
 * descriptor SATD kernel), "qr8_tile_kernel" (0: 8x8 TUs of the fused quantize_residual on the register kernel instead of sixteen to a
 * matrix-core tile), "qr_tile_pipe" (0: the tile kernels' wait on vector memory left to the compiler instead of placed before the
 * iteration's first store), "pipe" / "dct_pipe" (1: the same hand placement in the 4x4 / 8x8 register kernels / the 32x32
 * transform, where it measured slower), "full_qsad" (0: the search service's exhaustive search prices positions with v_sad_u8 on
 * byte-aligned operands instead of four alignments per v_qsad_pk_u16_u8), "sample8_wave" (0: 8x8 luma blocks of the sampling entry
 * on the general path), "service_workers" (resident workgroups of the search service, 0: a launch per batch), "service_push" (0: the workers read the units from
 * host memory even where the host could write them into device memory through a large BAR), "service_linger_us" /
 * "service_life_ms" (how long they stay without work / at most), "service_inflight" / "service_streams" (batches in the air and
 * launch streams of the launch-per-batch way), "service_spin_us" / "service_spin_crowded_us" / "service_nap_us" (how long a caller polls for its answer
 * before it naps: 100 us with a core per caller, 5 us and naps of 10 us with more callers than cores), "service_ticket_base_k" (tests: first ticket of the ring x 1024); the service reads its knobs when it is created.  KVZ_HIP_SERVICE_DEBUG in the
 * environment: a few lines of the workers' own timing on stderr when a service is destroyed.
 * Returns KVZ_HIP_OK or KVZ_HIP_ERR_INVALID (unknown key).  The environment variable KVZ_HIP_TUNE="key=value,..." presets
 * knobs at kvz_hip_init() for A/B runs of an unmodified host. */
KVZ_HIP_API int kvz_hip_set_tuning(const char *key, int value);

/* Thin device-memory helpers so a C host needs no HIP headers. */
KVZ_HIP_API void *kvz_hip_malloc(size_t bytes);
KVZ_HIP_API void kvz_hip_free(void *dptr);
/* page-locked host memory: what kvz_hip_memcpy_h2d / _d2h move without an intermediate copy (a host that feeds the
 * batched entries front by front -- descriptors up, results down -- stages them here) */
KVZ_HIP_API void *kvz_hip_malloc_host(size_t bytes);
KVZ_HIP_API void kvz_hip_free_host(void *hptr);
KVZ_HIP_API int kvz_hip_memcpy_h2d(void *dst, const void *src, size_t bytes, kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_memcpy_d2h(void *dst, const void *src, size_t bytes, kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_memset(void *dst, int value, size_t bytes, kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_memcpy_d2d(void *dst, const void *src, size_t bytes, kvz_hip_stream s);
/* The reconstructed-pixel exchange at a shard boundary (the reference's analogue is the bounded
 * cross-row read of WPP / tiles, encoderstate.c:777-828, encoder.c:240-241) for a host that drives several devices
 * from one process: an asynchronous peer copy (xGMI) of `bytes` from src on src_device to dst on dst_device, on a
 * stream of the calling thread's current device.  Both devices must have been initialised.  A CTU-row shard sends
 * the `margin` rows at its top / bottom edge into the halo rows of the shard above / below: with stride == width
 * those rows are one contiguous range, i.e. one call per direction per plane. */
KVZ_HIP_API int kvz_hip_memcpy_peer(void *dst, int dst_device, const void *src, int src_device, size_t bytes, kvz_hip_stream s);
/* The calling thread's current device must be src_device or dst_device; peer access towards the other one is enabled on first
 * use (an error if the devices cannot reach each other); a stream that belongs to another device is refused.
 *
 * One shard's whole exchange step in one call: the C form of kvazaar_amd/shard.py exchange_halo_into for a host whose shards
 * live on several devices of ONE process.  A shard's plane is an EXTENDED buffer: `rows` own rows starting at row `top`, the
 * neighbour's rows (the halo, `margin` of them) above and below.  The calling thread's shard PUSHES the first `margin` of its own
 * rows into the halo below the own rows of `up`, and its last `margin` own rows into the halo above the own rows of `down`
 * (either may be NULL: the frame's edge).  Asynchronous on stream s of self->device; the receiving shard orders its search
 * behind it with kvz_hip_event_record on that stream + kvz_hip_stream_wait_event on its own. */
typedef struct {
  void *ext;                    /* the shard's extended plane on `device`: (top + rows + halo below) rows of `stride` bytes */
  int32_t device;
  int32_t top, rows;            /* where the own rows start, and how many there are */
} kvz_hip_shard_plane;
KVZ_HIP_API int kvz_hip_halo_exchange(const kvz_hip_shard_plane *self, const kvz_hip_shard_plane *up, const kvz_hip_shard_plane *down,
                                      uint32_t stride, int margin, kvz_hip_stream s);
KVZ_HIP_API kvz_hip_stream kvz_hip_stream_create(void);
KVZ_HIP_API void kvz_hip_stream_destroy(kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_stream_sync(kvz_hip_stream s);
/* HIP-event timing of the enqueued work on stream s (used by bench.py for the
 * roofline: the event pair brackets exactly the launches issued in between). */
KVZ_HIP_API void *kvz_hip_event_create(void);
KVZ_HIP_API void kvz_hip_event_destroy(void *ev);
KVZ_HIP_API int kvz_hip_event_record(void *ev, kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_event_elapsed_ms(void *start, void *stop, float *ms);  /* syncs on stop */

/* A frame's launch sequence as one hipGraph (the encoder issues the same batched entries, on
 * the same buffers, for every frame: encoderstate.c:1330-1400 walks the same LCU grid per frame).
 * Every batched entry above and below is capture-safe: it only enqueues kernels on `s`.
 *   kvz_hip_graph_begin(s);  ...batched entries on s (and on streams forked from it with
 *   kvz_hip_event_record + kvz_hip_stream_wait_event, joined back the same way)...
 *   kvz_hip_graph_end(s, &g);  then per frame: kvz_hip_graph_launch(g, s).
 * Independent stages (motion search, intra search, SAO statistics) put on forked streams become
 * parallel branches of the graph and share the 256 CUs when one stage alone cannot fill them. */
typedef void *kvz_hip_graph;         /* an instantiated graph (hipGraphExec_t) */
KVZ_HIP_API int kvz_hip_stream_wait_event(kvz_hip_stream s, void *ev);
KVZ_HIP_API int kvz_hip_graph_begin(kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_graph_end(kvz_hip_stream s, kvz_hip_graph *graph_out);
KVZ_HIP_API int kvz_hip_graph_launch(kvz_hip_graph graph, kvz_hip_stream s);
KVZ_HIP_API void kvz_hip_graph_destroy(kvz_hip_graph graph);

/* ------------------------------------------------------------------ */
/* (2) batched entries -- picture group                               */
/*     reference typedefs: strategies/strategies-picture.h:102-130    */
/* ------------------------------------------------------------------ */

/* cost_pixel_nxn_func over `count` contiguous N*N block pairs:
 * costs[i] = sad_NxN(blk1 + i*N*N, blk2 + i*N*N); N in {4,8,16,32,64}.
 * Replaces sad_4x4..sad_64x64 (generic/picture-generic.c:460-486). */
KVZ_HIP_API int kvz_hip_sad_nxn_batch(int n, const kvz_hip_pixel *blk1, const kvz_hip_pixel *blk2,
                                      size_t count, uint32_t *costs, kvz_hip_stream s);
/* satd_4x4..satd_64x64 (picture-generic.c:189-196, strategies-picture.h:40-56) */
KVZ_HIP_API int kvz_hip_satd_nxn_batch(int n, const kvz_hip_pixel *blk1, const kvz_hip_pixel *blk2,
                                       size_t count, uint32_t *costs, kvz_hip_stream s);

/* cost_pixel_nxn_multi_func (sad_NxN_dual / satd_NxN_dual, picture-generic.c:357-390,
 * :497-519): item i has two predictions at preds + i*item_stride + {0, pred_stride}
 * (pred_stride = 1024 for the reference's pred_buffer, strategies-picture.h:34) and
 * one original block at orig + i*N*N; costs[2*i + k]. */
KVZ_HIP_API int kvz_hip_sad_nxn_dual_batch(int n, const kvz_hip_pixel *preds, size_t pred_stride, size_t item_stride,
                                           const kvz_hip_pixel *orig, size_t count, uint32_t *costs, kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_satd_nxn_dual_batch(int n, const kvz_hip_pixel *preds, size_t pred_stride, size_t item_stride,
                                            const kvz_hip_pixel *orig, size_t count, uint32_t *costs, kvz_hip_stream s);

/* One block pair inside two planes; the unit of the frame-level entries below. */
typedef struct {
  int32_t x1, y1;          /* top-left of the block in plane 1 (the picture being coded) */
  int32_t x2, y2;          /* top-left of the block in plane 2 (the reference picture)   */
  int32_t width, height;
} kvz_hip_block_pair;

/* reg_sad_func (picture-generic.c:86-99) over a list of block pairs that lie
 * INSIDE their planes: costs[i] = reg_sad(p1 + y1*stride1 + x1, p2 + y2*stride2 + x2, w, h, ..).
 * Any width/height >= 1 (tests/sad_tests.c:369-376 shapes, 64x63, 1x1). */
KVZ_HIP_API int kvz_hip_reg_sad_batch(const kvz_hip_pixel *plane1, uint32_t stride1,
                                      const kvz_hip_pixel *plane2, uint32_t stride2,
                                      const kvz_hip_block_pair *pairs, size_t count,
                                      uint32_t *costs, kvz_hip_stream s);
/* kvz_image_calc_sad (image.c:455-486): plane 2 coordinates may be outside the
 * w2 x h2 reference frame; outside pixels are edge replicated, which is what
 * image_interpolated_sad (image.c:320-444) computes. */
KVZ_HIP_API int kvz_hip_image_calc_sad_batch(const kvz_hip_pixel *pic, uint32_t pic_stride,
                                             const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                             const kvz_hip_block_pair *pairs, size_t count,
                                             uint32_t *costs, kvz_hip_stream s);
/* cost_pixel_any_size_func (strategies-picture.h:62-100) / kvz_image_calc_satd
 * (image.c:488-545): width and height multiples of 4; edge replication as above. */
KVZ_HIP_API int kvz_hip_image_calc_satd_batch(const kvz_hip_pixel *pic, uint32_t pic_stride,
                                              const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                              const kvz_hip_block_pair *pairs, size_t count,
                                              uint32_t *costs, kvz_hip_stream s);
/* pixels_calc_ssd_func (picture-generic.c:521-536): square blocks, width = pairs[i].width */
KVZ_HIP_API int kvz_hip_pixels_calc_ssd_batch(const kvz_hip_pixel *plane1, uint32_t stride1,
                                              const kvz_hip_pixel *plane2, uint32_t stride2,
                                              const kvz_hip_block_pair *pairs, size_t count,
                                              uint32_t *ssd, kvz_hip_stream s);
/* cost_pixel_any_size_multi_func (satd_any_size_quad, picture-generic.c:392-456),
 * including its behaviour for widths/heights that are not multiples of 8:
 * item i compares 4 candidate blocks preds + (4*i + k)*pred_item_stride
 * (row stride pred_stride, 64 in search_inter.c:1076) with the block at
 * (x1,y1) of `orig`; costs[4*i + k]. */
KVZ_HIP_API int kvz_hip_satd_any_size_quad_batch(const kvz_hip_pixel *preds, uint32_t pred_stride, size_t pred_item_stride,
                                                 const kvz_hip_pixel *orig, uint32_t orig_stride,
                                                 const kvz_hip_block_pair *pairs, size_t count,
                                                 uint32_t *costs, kvz_hip_stream s);
/* Batched integer motion-estimation costs, one CTU per workgroup: what check_mv_cost
 * (search_inter.c:195-232) computes one candidate at a time through
 * kvz_image_calc_sad (image.c:455-486), for every candidate of a pattern and every
 * square PU size of the CTU in one launch.  For CTU i (64x64 at (x, y) in `pic`),
 * search centre (mvx, mvy) in full pels, and candidate m = centre + mv_offsets[m]:
 *   costs[(i*n_mv + m)*85 + k] = kvz_image_calc_sad(pic, ref, bx, by, bx + mv, by + mv, bw, bw)
 * with k = 0: the 64x64 PU; 1..4: the 32x32 PUs, 5..20: 16x16, 21..84: 8x8, each
 * group in raster order inside the CTU.  The reference block may leave the frame
 * (edge replicated).  PUs not entirely inside `pic` (ragged last CTU row/column)
 * get 0xFFFFFFFF.  The CTU's source block and its search window are fetched from
 * HBM once and stay in LDS; larger PUs are sums of the 8x8 SADs.
 * |mv_offsets| <= 64 in each component. */
typedef struct {
  int32_t x, y;            /* CTU top-left in the picture (multiples of 8) */
  int32_t mvx, mvy;        /* search centre, full-pel */
} kvz_hip_ctu_search;
#define KVZ_HIP_CTU_PUS 85
KVZ_HIP_API int kvz_hip_ctu_sad_grid_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, int pic_w, int pic_h,
                                           const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                           const kvz_hip_ctu_search *ctus, size_t count,
                                           const int16_t *mv_offsets /* device, [n_mv][2] = (dx, dy) */, int n_mv,
                                           uint32_t *costs /* [count][n_mv][85] */, kvz_hip_stream s);

/* inter_recon_bipred_func's blend (picture-generic.c:538-588) for `count` planes
 * of w x h samples laid out contiguously (stride w): each source is either 14-bit
 * int16 samples (hi_prec != 0) or pixels. */
KVZ_HIP_API int kvz_hip_bipred_blend_batch(int w, int h, int hi_prec0, const void *src0, int hi_prec1, const void *src1,
                                           kvz_hip_pixel *dst, size_t count, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* (2) batched entries -- dct group (strategies/strategies-dct.h:31)   */
/* ------------------------------------------------------------------ */
enum { KVZ_HIP_DCT = 0, KVZ_HIP_IDCT = 1, KVZ_HIP_DST = 2, KVZ_HIP_IDST = 3, KVZ_HIP_TRSKIP = 4, KVZ_HIP_ITRSKIP = 5 };
/* dct_func over `count` contiguous N*N int16 blocks (generic/dct-generic.c:567-617).
 * kind DCT/IDCT: n in {4,8,16,32}; DST/IDST: n == 4 (fast_forward_dst_4x4 / inverse);
 * TRSKIP/ITRSKIP: kvz_transformskip / kvz_itransformskip (transform.c:150-180), n in {4,8,16,32}. */
KVZ_HIP_API int kvz_hip_transform_batch(int kind, int n, const int16_t *in, int16_t *out,
                                        size_t count, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* (2) batched entries -- quant group (strategies/strategies-quant.h)  */
/* ------------------------------------------------------------------ */
/* The encoder state the reference's quant functions read, flattened
 * (generic/quant-generic.c:40-50, :283-289). */
typedef struct {
  int32_t qp;               /* state->qp */
  int32_t slice_is_intra;   /* state->frame->slicetype == KVZ_SLICE_I */
  int32_t signhide;         /* encoder->cfg.signhide_enable */
  int32_t scaling_list;     /* encoder->scaling_list.enable; 0 = flat list */
  const int32_t *quant_coeff;    /* DEVICE (batched) / HOST (per-call) [w*h] factors when scaling_list */
  const int32_t *dequant_coeff;  /* likewise */
} kvz_hip_quant_params;

/* quant_func (quant-generic.c:37-163) over `count` w*w blocks.
 * type: 0 luma, 2/3 chroma; scan_idx 0 diag / 1 hor / 2 ver. */
KVZ_HIP_API int kvz_hip_quant_batch(const kvz_hip_quant_params *p, const kvz_hip_coeff *coef, kvz_hip_coeff *q_coef,
                                    int width, int type, int scan_idx, size_t count, kvz_hip_stream s);
/* dequant_func (quant-generic.c:279-321) */
KVZ_HIP_API int kvz_hip_dequant_batch(const kvz_hip_quant_params *p, const kvz_hip_coeff *q_coef, kvz_hip_coeff *coef,
                                      int width, int type, size_t count, kvz_hip_stream s);
/* coeff_abs_sum_func (quant-generic.c:323-330) per block of `length` coeffs */
KVZ_HIP_API int kvz_hip_coeff_abs_sum_batch(const kvz_hip_coeff *coeffs, size_t length, size_t count,
                                            uint32_t *sums, kvz_hip_stream s);
/* quant_residual_func, the rdoq-off path of kvz_quantize_residual_generic
 * (quant-generic.c:180-273): residual -> transform -> quant -> dequant ->
 * inverse -> reconstruction, one fused kernel per TU.  Blocks are contiguous
 * (stride = width).  rec_out may alias pred_in.  has_coeffs[i] receives the
 * function's return value.  color 0 Y / 1 U / 2 V. */
KVZ_HIP_API int kvz_hip_quantize_residual_batch(const kvz_hip_quant_params *p, int cu_is_intra, int width, int color,
                                                int scan_order, int use_trskip,
                                                const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in,
                                                kvz_hip_pixel *rec_out, kvz_hip_coeff *coeff_out, int32_t *has_coeffs,
                                                size_t count, kvz_hip_stream s);
/* The two elementwise ends of kvz_quantize_residual as separate entries (quant-generic.c:196-204 and :253-259), for
 * callers that run something of their own between transform and dequantisation (the drop-in does, when RDOQ is on):
 * residual[i] = ref[i] - pred[i];  rec[i] = clip((int16)(residual[i] + pred[i])).  n = number of pixels. */
KVZ_HIP_API int kvz_hip_residual_batch(const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in, kvz_hip_coeff *residual,
                                       size_t n, kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_reconstruct_batch(const kvz_hip_coeff *residual, const kvz_hip_pixel *pred_in, kvz_hip_pixel *rec_out,
                                          size_t n, kvz_hip_stream s);
/* The same fused kernel with the two numbers the rd=0 TU cost is made of
 * (kvz_cu_rd_cost_luma / _chroma, search.c:236-306, :309-383): ssd_out[i] =
 * kvz_pixels_calc_ssd(ref_i, rec_i, width) (picture-generic.c:521-536) and
 * coeff_abs_sum_out[i] = kvz_coeff_abs_sum(coeff_i, width*width) (rdo.c:219,
 * quant-generic.c:323-330), taken from the registers / LDS the reconstruction
 * and coefficients are in -- neither array is read back from HBM.
 * SURVEY.md section 8(f) row 3. */
KVZ_HIP_API int kvz_hip_quantize_residual_cost_batch(const kvz_hip_quant_params *p, int cu_is_intra, int width, int color,
                                                     int scan_order, int use_trskip,
                                                     const kvz_hip_pixel *ref_in, const kvz_hip_pixel *pred_in,
                                                     kvz_hip_pixel *rec_out, kvz_hip_coeff *coeff_out, int32_t *has_coeffs,
                                                     uint32_t *ssd_out, uint32_t *coeff_abs_sum_out,
                                                     size_t count, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* (2) batched entries -- ipol group (strategies/strategies-ipol.h)    */
/* ------------------------------------------------------------------ */
typedef struct {
  int32_t x, y;            /* integer-pel top-left of the block in the reference plane (may be outside) */
  int32_t mv_frac_x, mv_frac_y;   /* mv & 3 (luma) / mv & 7 (chroma) */
  int32_t width, height;
} kvz_hip_ipol_block;

/* kvz_sample_quarterpel_luma / kvz_sample_octpel_chroma and their 14-bit
 * variants (generic/ipol-generic.c:122-190, :660-728), with the source window
 * fetched like kvz_get_extended_block (ipol-generic.c:731-784: coordinates
 * clamped to the plane).  Output block i is written contiguously (stride =
 * width) at dst + out_offsets[i] (elements). */
KVZ_HIP_API int kvz_hip_sample_luma_batch(const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                          const kvz_hip_ipol_block *blocks, const uint64_t *out_offsets, size_t count,
                                          int out_14bit, void *dst, kvz_hip_stream s);
KVZ_HIP_API int kvz_hip_sample_chroma_batch(const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                            const kvz_hip_ipol_block *blocks, const uint64_t *out_offsets, size_t count,
                                            int out_14bit, void *dst, kvz_hip_stream s);

/* Fractional motion search of search_frac (search_inter.c:965-1128): for block
 * pair i (x1,y1 in pic; x2,y2 = integer-pel position in ref; w,h multiples of
 * 4 up to 64, not both 4 mod 8 -- for the SMP / AMP shapes the integer position
 * is scored by satd_any_size and the candidates by satd_any_size_quad, whose
 * 4x4 stages add nothing, exactly as the reference does) the four filter steps (filter_hpel/qpel_blocks_*_luma,
 * ipol-generic.c:192-658) fused with satd_any_size(_quad); filtered candidates
 * stay in LDS, only costs leave the CU.  costs[17*i + 0] integer position,
 * [1..8] half-pel neighbours, [9..16] quarter-pel neighbours of the best
 * half-pel position; best[2*i + {0,1}] = chosen hpel / qpel index (0 = centre),
 * ties and order as in the reference, MV bit costs taken as zero. */
KVZ_HIP_API int kvz_hip_search_frac_batch(const kvz_hip_pixel *pic, uint32_t pic_stride,
                                          const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                          const kvz_hip_block_pair *pairs, size_t count,
                                          uint32_t *costs, int32_t *best, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* (2) batched entries -- motion search of whole PUs                   */
/*     SURVEY.md section 8(f) row 1                                    */
/* ------------------------------------------------------------------ */
/* One merge candidate as calc_mvd_cost / select_starting_point see it
 * (inter_merge_cand_t, inter.h): */
typedef struct {
  int16_t mv[2];       /* merge_cand[i].mv[dir - 1], quarter-pel */
  uint8_t usable;      /* merge_cand[i].dir != 3 */
  uint8_t same_ref;    /* state->frame->ref_LX[dir - 1][merge_cand[i].ref[dir - 1]] == ref_idx */
} kvz_hip_me_merge;
/* inter_search_info_t (search_inter.c:40-76) of one PU for one reference picture */
typedef struct {
  int32_t x, y, width, height;   /* PU inside the picture; width, height multiples of 4 in 4..64, not both 4 mod 8:
                                  * every shape of the inter search incl. the SMP / AMP ones (8x4, 4x8, 16x4, 4x16, 16x12, 12x16) */
  int16_t mv_cand[2][2];         /* AMVP candidates (kvz_inter_get_mv_cand), quarter-pel */
  int16_t extra_mv[2];           /* start vector from the co-located CU (search_inter.c:1190-1206), quarter-pel */
  int16_t num_merge_cand;        /* 0..5 */
  int16_t reserved;              /* mv_rdo: index of this PU's snapshot in kvz_hip_me_params.cabac */
  kvz_hip_me_merge merge[5];
  int16_t pad;                   /* bits 0-1: merge neighbours barred (kvz_hip_inter_candidates_batch); bits 2..: plane pair
                                    (kvz_hip_search_pu_multi_batch); ignored by kvz_hip_search_pu_batch */
} kvz_hip_me_pu;                 /* 64 bytes */
/* The CABAC state kvz_calc_mvd_cost_cabac (rdo.c:908-1060, --mv-rdo) starts from: what it reads of state->cabac
 * (cabac_data_t, cabac.h:41-88).  The encoder's contexts change from LCU to LCU; a batch carries one snapshot per
 * distinct state and every PU names its own (kvz_hip_me_pu.reserved). */
typedef struct {
  uint16_t range;                /* cabac.range */
  uint8_t ctx[8];                /* uc_state of ctx.cu_merge_flag_ext_model, cu_merge_idx_ext_model, cu_ref_pic_model[0], [1],
                                    cu_mvd_model[0], [1], mvp_idx_model[0]; [7] unused */
  uint8_t pad[6];
} kvz_hip_me_cabac;              /* 16 bytes */
/* the encoder settings the search reads */
typedef struct {
  int32_t lambda_cost;           /* (int32_t)(state->lambda_sqrt + 0.5), search_inter.c:411 */
  int32_t early_termination;     /* cfg.me_early_termination: 0 off, 1 on, 2 sensitive */
  uint32_t max_steps;            /* cfg.me_max_steps */
  int32_t fme_level;             /* cfg.fme_level, 0..4 */
  int32_t wpp_owf;               /* cfg.owf && cfg.wpp: enforce the reference-availability rule of fracmv_within_tile */
  int32_t ref_delay_px;          /* SAO_DELAY_PX (sao on), DEBLOCK_DELAY_PX (deblock only) or 0 (global.h:163,175) */
  int32_t max_ref_lcu_down, max_ref_lcu_right;   /* ctrl->max_inter_ref_lcu (encoder.c:240-241) */
  int32_t algorithm;             /* cfg.ime_algorithm: 0 hexbs (hexagon_search), 1 dia (diamond_search, :796-883), 2 tz (tz_search, :595-672),
                                    3 full (search_mv_full, :886-962; no early termination) */
  int32_t search_range;          /* algorithm 3 only: 8, 16, 32 or 64 (search_inter.c:1208-1215), any value 1..64 accepted */
  int32_t size_classes;          /* optional hint, 0 = unknown: OR of 1 (PUs up to 16x16), 2 (up to 32x32), 4 (larger) present in the
                                    batch; the entry runs one kernel per size class and skips the launches for absent ones */
  int32_t mv_constraint;         /* cfg.mv_constraint (enum kvz_mv_constraint, kvazaar.h:113-119): 0 none, 1 frame, 2 tile, 3 frame and tile,
                                    4 frame and tile with the interpolation margin -- the five branches of fracmv_within_tile
                                    (search_inter.c:142-171; the reference treats 1..3 alike) */
  int32_t tile_x, tile_y;        /* state->tile->offset_x / offset_y: top-left of the tile in the picture (multiples of 64 in the reference;
                                    required here only with wpp_owf, whose rule counts LCUs from the tile origin) */
  int32_t tile_w, tile_h;        /* state->tile->frame->width / height; 0 x 0 = the whole picture is one tile.  PU coordinates stay
                                    picture coordinates; the entry derives the tile-relative info->origin the reference tests.
                                    A CTU-row shard of a frame is a tile of full width: with mv_constraint 3 or 4 its
                                    search reads nothing outside its own rows (+ nothing at all beyond them), with wpp_owf and
                                    max_ref_lcu_down = 1 nothing beyond one CTU row + ref_delay_px + 4 rows below them. */
  int32_t mv_rdo;                /* cfg.mv_rdo: MV bit costs from the CABAC model (kvz_calc_mvd_cost_cabac, rdo.c:908-1060, and
                                    kvz_get_mvd_coding_cost_cabac in select_mv_cand) instead of the exp-Golomb estimate */
  int32_t ref_idx;               /* mv_rdo: info->ref_idx of the reference picture searched */
  int32_t refs_before;           /* mv_rdo: pictures of state->frame->ref with poc < current poc (rdo.c:990-998); ref_idx is coded when > 1 */
  int32_t n_cabac;               /* mv_rdo: number of snapshots in cabac (>= 1); a PU whose kvz_hip_me_pu.reserved is not an index into it is
                                    flagged (cost 0xFFFFFFFF, reserved -1) instead of searched.  Else unused (0) */
  const kvz_hip_me_cabac *cabac; /* mv_rdo: DEVICE array of n_cabac snapshots, indexed by kvz_hip_me_pu.reserved; else unused (NULL) */
  const uint32_t *cost_to_beat;  /* NULL, or a DEVICE array with one entry per PU: *inter_cost as search_pu_inter_ref finds it
                                    (search_inter.c:1239), i.e. the best cost of the reference pictures searched before this one
                                    (MAX_INT for the first).  A PU whose integer search does not get below it skips the fractional
                                    search and is scored like fme_level 0 (:1242-1252), exactly as the reference's loop over
                                    the pictures of a multi-reference frame does */
} kvz_hip_me_params;             /* 96 bytes */
typedef struct {
  int32_t mv[2];                 /* info->best_mv, quarter-pel */
  uint32_t cost, bitcost;        /* info->best_cost, info->best_bitcost; cost 0xFFFFFFFF: nothing allowed / bad descriptor */
  int32_t merged, merge_idx;     /* the match loop of search_inter.c:1253-1266 */
  int32_t mv_cand;               /* select_mv_cand (search_inter.c:1268-1273) */
  int32_t reserved;              /* -1 flags a malformed descriptor */
} kvz_hip_me_result;

/* The --me hexbs / dia / tz / full paths of search_pu_inter_ref (search_inter.c:1134-1300) for
 * `count` PUs against one reference plane in one launch: hexagon_search
 * diamond_search or tz_search (:463-883, with select_starting_point and early_terminate) over
 * kvz_image_calc_sad + calc_mvd_cost, then search_frac (:965-1128) -- or, for
 * fme_level 0, the SATD re-cost of :1236-1248.  Same visiting order and
 * tie-breaks as the reference, so results[i] equals what the reference leaves
 * in inter_search_info_t.  pus / results are device arrays. */
KVZ_HIP_API int kvz_hip_search_pu_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, int pic_w, int pic_h,
                                        const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                        const kvz_hip_me_pu *pus, size_t count, const kvz_hip_me_params *params,
                                        kvz_hip_me_result *results, kvz_hip_stream s);
/* The same search over SEVERAL pictures of one size in one launch: pics / refs are DEVICE arrays of n_planes plane pointers
 * (all planes share the strides and sizes given), and a PU names its pair in kvz_hip_me_pu.pad >> 2 (bits 0 and 1 keep
 * their meaning for kvz_hip_inter_candidates_batch).  This is how work that is independent by construction -- the same
 * dependency front of several frames in flight (--owf), of several tiles, of several encoder instances -- shares a launch:
 * a front of a dozen PUs leaves the chip idle, and host threads stop scaling at the runtime's launch rate, but fronts of
 * many pictures merged into one launch cost what one does.  A plane index outside 0 .. n_planes - 1 flags the PU
 * (reserved -1).  mv_rdo is not available here. */
KVZ_HIP_API int kvz_hip_search_pu_multi_batch(const kvz_hip_pixel *const *pics, uint32_t pic_stride, int pic_w, int pic_h,
                                              const kvz_hip_pixel *const *refs, uint32_t ref_stride, int ref_w, int ref_h, int n_planes,
                                              const kvz_hip_me_pu *pus, size_t count, const kvz_hip_me_params *params,
                                              kvz_hip_me_result *results, kvz_hip_stream s);

/* ---- search service: the searches of MANY host threads, without a launch per search ---- */
/* The reference runs one CTU job per threadqueue worker (encoderstate.c:777-828: WPP rows, frames in flight under
 * --owf, tiles), and every worker reaches search_pu_inter (search_inter.c:1451-1520) with ONE PU at a time.  A launch per
 * PU and per reference picture leaves the chip idle and pays the launch price every time; the service is the piece
 * between those workers and the kernels of kvz_hip_search_pu_batch:
 *   - the luma planes the searches read (source pictures, reconstructed pictures) stay resident on the device in
 *     numbered slots, written rectangle by rectangle as the host produces them;
 *   - kvz_hip_me_service_search() is called concurrently by the workers, each with one PU and ALL the reference pictures
 *     of search_pu_inter's loop (:1502-1507).  By default the (PU, picture) units go into a ring in page-locked memory and are
 *     taken by workgroups that stay on the device ("resident workers": started when a request finds none, gone when the
 *     service has been idle for a couple of milliseconds; tuning "service_workers" = their number, default 64, 0 = off).
 *     Without them -- and always with more than 128 calling threads -- requests that are pending at the same time leave in
 *     one launch: whichever caller finds the launch path free takes everything that is queued, so batches grow with the load;
 *     at most "service_inflight" (4) such batches are in the air;
 *   - the reference pictures of a PU are searched IN PARALLEL.  The loop of :1502-1507 is sequential only through
 *     *inter_cost: a picture whose integer search does not get below it skips search_frac and is re-scored with SATD
 *     (:1239-1252).  Every (PU, picture) unit therefore computes both outcomes -- the search with its fractional stage,
 *     and the integer vector with the SATD cost, which is search_frac's own first candidate (:1019-1029) -- and the call
 *     replays the sequential rule over the units before it returns.  results[i] is what search_pu_inter_ref leaves in
 *     info for picture i when the pictures are visited in order, starting from cost_to_beat;
 *   - descriptors are read, and results written, by the kernels in page-locked host memory: no copy commands, a caller
 *     waits on its own results only.
 * One service per device and picture size.  Thread-safe.  mv_rdo is not available here. */
typedef struct kvz_hip_me_service kvz_hip_me_service;
#define KVZ_HIP_SERVICE_MAX_REFS 16
typedef struct {
  int32_t width, height;        /* luma size of every picture of the service */
  int32_t max_pictures;         /* slots, 1..256 */
  int32_t max_threads;          /* host threads that will ever call kvz_hip_me_service_search (each gets its own result area), 1..1024 */
  int32_t reserved[4];          /* 0 */
} kvz_hip_me_service_config;
typedef struct {
  int32_t pic_slot;             /* the picture being coded */
  int32_t n_refs;               /* 1..KVZ_HIP_SERVICE_MAX_REFS */
  int32_t ref_slot[KVZ_HIP_SERVICE_MAX_REFS];
  uint32_t cost_to_beat;        /* *inter_cost as search_pu_inter hands it to the first picture (MAX_INT, :1492) */
  int32_t reserved;
  kvz_hip_me_params params;     /* as for kvz_hip_search_pu_batch; cost_to_beat / cabac / mv_rdo / size_classes unused (NULL, 0) */
  kvz_hip_me_pu pu[KVZ_HIP_SERVICE_MAX_REFS];   /* the PU as picture i sees it: mv_cand, extra_mv and merge[].same_ref differ per picture */
} kvz_hip_me_request;
typedef struct {
  uint64_t requests, units;     /* searches asked for; (PU, picture) units searched */
  uint64_t batches, launches;   /* a launch per batch: times the queue was drained = kernel launches; resident workers: requests posted to the ring, times workers were started */
  uint64_t max_batch_units;     /* (0 with resident workers) */
  uint64_t rects, rect_bytes;   /* kvz_hip_me_service_put_rect calls and the bytes they moved */
  uint64_t wait_ns;             /* summed over callers: time between posting a request and seeing its results */
  uint64_t tables, table_bytes; /* kvz_hip_me_service_sad_tables: (CTU, picture) tables filled and the bytes the kernels wrote to host memory */
  uint64_t table_ns;            /* summed over callers: time spent in kvz_hip_me_service_sad_tables */
} kvz_hip_me_service_stats;
KVZ_HIP_API kvz_hip_me_service *kvz_hip_me_service_create(const kvz_hip_me_service_config *cfg);
KVZ_HIP_API void kvz_hip_me_service_destroy(kvz_hip_me_service *svc);
/* Copies a w x h rectangle of a host plane (host points at its top-left pixel) to (x, y) of slot's plane; complete on
 * return, so a request posted afterwards by any thread reads it. */
KVZ_HIP_API int kvz_hip_me_service_put_rect(kvz_hip_me_service *svc, int slot, const kvz_hip_pixel *host, uint32_t host_stride,
                                            int x, int y, int w, int h);
/* Blocks until the request's results[0 .. n_refs - 1] are there.  KVZ_HIP_ERR_INVALID for a malformed request (slots, n_refs, parameters,
 * mv_rdo; with the exhaustive search a pu.x that is not a multiple of 4) or PU (outside the picture, impossible shape). */
KVZ_HIP_API int kvz_hip_me_service_search(kvz_hip_me_service *svc, const kvz_hip_me_request *req, kvz_hip_me_result *results);
KVZ_HIP_API int kvz_hip_me_service_get_stats(kvz_hip_me_service *svc, kvz_hip_me_service_stats *out);
/* The device plane of a slot, for a host that wants to run another batched entry on the resident pictures. */
KVZ_HIP_API const kvz_hip_pixel *kvz_hip_me_service_plane(kvz_hip_me_service *svc, int slot);
/* The candidate-INDEPENDENT half of the integer search, for hosts that keep the search itself: every value check_mv_cost
 * (search_inter.c:195-232) can ask kvz_image_calc_sad (image.c:455-486) for while it searches the square PUs of one CTU
 * within +-range full pixels of their own position -- no neighbour decision enters, so the host can ask for it when it
 * starts the CTU.  One kvz_hip_ctu_sad_grid_batch launch per picture (a workgroup per window row: the CTU's source block and
 * that row's reference pixels staged in LDS once, larger PUs as sums of the 8x8 SADs), written by the kernels straight into
 * page-locked host memory owned by the service.  Returns the calling thread's table, valid until its next call:
 *   table[((i * side + (dy + range)) * side + (dx + range)) * KVZ_HIP_CTU_PUS + k],  side = 2 * range + 1,
 * i = index in ref_slots, k = the PU as in kvz_hip_ctu_sad_grid_batch (0: 64x64; 1..4: 32x32; 5..20: 16x16; 21..84: 8x8, raster
 * order inside the CTU); 0xFFFFFFFF = the PU is not inside the picture.  Reference blocks that leave the picture are edge
 * replicated like image_interpolated_sad (image.c:320-444).  range 1..32; ctu_x, ctu_y multiples of 64.  NULL on failure.
 * Bytes per call: n_refs * side^2 * 340 (range 16: 370 KB per picture). */
KVZ_HIP_API const uint32_t *kvz_hip_me_service_sad_tables(kvz_hip_me_service *svc, int pic_slot, int n_refs, const int32_t *ref_slots,
                                                          int ctu_x, int ctu_y, int range);

/* ---- candidate derivation next to the search: what a host derives between two dependency fronts ---- */
/* One record per 4x4 SCU, row-major: the fields of cu_info_t (cu.h:117-153) the candidate derivation and the
 * deblocking filter (below) read.  Inter CUs carry mv_dir 1..3. */
typedef struct {
  uint8_t type;                 /* cu_type_t: CU_INTRA 1, CU_INTER 2 (cu.h:38-43) */
  uint8_t depth, part_size, tr_depth;
  uint8_t cbf_y;                /* cbf_is_set(cu->cbf, cu->tr_depth, COLOR_Y) (cu.h:504-507) */
  uint8_t mv_dir;               /* inter.mv_dir */
  uint8_t qp;                   /* cu->qp */
  uint8_t reserved;
  int16_t mv[2][2];             /* inter.mv */
  uint8_t mv_ref[2];            /* inter.mv_ref */
  uint8_t pad[2];
} kvz_hip_cu_info;              /* 20 bytes */
/* What kvz_inter_get_mv_cand / kvz_inter_get_merge_cand (inter.c:1209-1446) read of the encoder state. */
typedef struct {
  int32_t poc;                   /* state->frame->poc */
  int32_t slice_is_b;            /* state->frame->slicetype == KVZ_SLICE_B */
  int32_t tmvp_enable;           /* cfg.tmvp_enable */
  int32_t num_refs;              /* state->frame->ref->used_size, 0..16 */
  int32_t ref_pocs[16];          /* state->frame->ref->pocs */
  uint8_t ref_LX[2][16];         /* state->frame->ref_LX (encoderstate.h:100) */
  uint8_t ref_LX_size[2];        /* state->frame->ref_LX_size */
  uint8_t pad[2];
  int32_t col_ref_pocs[16];      /* state->frame->ref->images[c]->ref_pocs, c = ref_LX[0][0]: the collocated picture (inter.c:1020-1055) */
  uint8_t col_ref_LX[2][16];     /* state->frame->ref->ref_LXs[c] */
  int32_t pic_width, pic_height; /* state->tile->frame->width / height: the (tile) picture the reference's functions see */
  int32_t in_width, in_height;   /* encoder_control->in.width / height: the bounds of the temporal neighbours (inter.c:747,763) */
  int32_t tile_x, tile_y;        /* state->tile->offset_x / _y.  Descriptors carry PICTURE coordinates (the search entry's
                                    convention); the derivation subtracts the offset, cus is indexed tile-relative, col_cus /
                                    ref_cus picture-wide (the start vector's lookup, search_inter.c:1193-1194, adds it back) */
  int32_t ref_idx;               /* info->ref_idx: the picture of state->frame->ref about to be searched */
  int32_t cus_stride;            /* records per row of cus */
  int32_t col_stride;            /* records per row of col_cus / ref_cus (cu_array_t: the picture width rounded up to whole LCUs, / 4) */
  int32_t reserved;
} kvz_hip_inter_params;          /* 252 bytes */
/* inter_merge_cand_t (inter.h:36-41) */
typedef struct {
  uint8_t dir;                   /* 1 L0, 2 L1, 3 both */
  uint8_t ref[2];                /* index in L0 / L1 */
  uint8_t pad;
  int16_t mv[2][2];
} kvz_hip_merge_cand;            /* 12 bytes */
/* Completes the search descriptors of `count` PUs on the device -- everything search_pu_inter and
 * search_pu_inter_ref derive before the search of picture params->ref_idx (search_inter.c:1470-1500, :1143-1206):
 *   in : pus[i].x, y (picture coordinates), width, height (any PU shape of the inter search) and pus[i].pad: bit 0 = merge candidate A1
 *        barred, bit 1 = B1 barred (the second PU of a two-PU CU, search_inter.c:1470-1475);
 *   out: pus[i].num_merge_cand and merge[] (kvz_inter_get_merge_cand, inter.c:1314-1446, seen as calc_mvd_cost sees
 *        it), mv_cand (kvz_inter_get_mv_cand, inter.c:1209-1240, for the list and index that hold ref_idx),
 *        extra_mv (the vector of the CU of picture ref_idx under the PU's centre, search_inter.c:1190-1206);
 *        merge_out (DEVICE, 5 per PU, may be NULL): the full inter_merge_cand_t list for the merge / skip evaluation.
 * cus = the current (tile) picture's CUs, one kvz_hip_cu_info per 4x4 SCU, holding what lcu->cu holds while its LCU
 * is searched: the decided neighbours (type 0 = not coded yet; of a record only type, mv_dir, mv, mv_ref are read);
 * col_cus = the CUs of the collocated picture ref_LX[0][0] (needed when tmvp_enable and num_refs > 0), ref_cus = those
 * of picture ref_idx (may be NULL: extra_mv 0), both whole-picture arrays.  params is a HOST struct.  A descriptor
 * outside the picture or off the 4-pixel grid gets num_merge_cand -1.  The results feed kvz_hip_search_pu_batch on
 * the same stream: with the CU arrays resident, a dependency front costs no host round trip for its candidates. */
KVZ_HIP_API int kvz_hip_inter_candidates_batch(const kvz_hip_cu_info *cus, const kvz_hip_cu_info *col_cus, const kvz_hip_cu_info *ref_cus,
                                               const kvz_hip_inter_params *params, kvz_hip_me_pu *pus, size_t count,
                                               kvz_hip_merge_cand *merge_out, kvz_hip_stream s);

/* The same derivation for PUs of SEVERAL pictures in one launch (cf. kvz_hip_search_pu_multi_batch): one record per picture
 * in DEVICE memory, a PU names its picture in kvz_hip_me_pu.pad >> 2.  What the one-picture entry refuses as a bad
 * argument (table sizes, strides, a missing collocated array) makes the PUs of that picture read num_merge_cand -1 here,
 * as does a picture index outside 0 .. n_pictures - 1. */
typedef struct {
  const kvz_hip_cu_info *cus, *col_cus, *ref_cus;   /* as the arguments of kvz_hip_inter_candidates_batch */
  kvz_hip_inter_params params;
  int32_t reserved;
} kvz_hip_inter_picture;         /* 280 bytes */
KVZ_HIP_API int kvz_hip_inter_candidates_multi_batch(const kvz_hip_inter_picture *pictures, int n_pictures, kvz_hip_me_pu *pus, size_t count,
                                                     kvz_hip_merge_cand *merge_out, kvz_hip_stream s);


/* Bi-prediction candidate cost of search_pu_inter_bipred (search_inter.c:1304-1440): for candidate i the luma of
 * kvz_inter_recon_bipred (inter.c:430-477; a 14-bit quarter-pel sample per reference when its vector is fractional,
 * else the edge-clamped pixels << 6, blended and clipped) scored with kvz_satd_any_size against the source block
 * (:1359-1362).  The caller adds the MV bit costs (:1366-1389).  Quarter-pel vectors; both reference planes have
 * ref_w x ref_h pixels; width, height multiples of 4 in 4..64 and not both 4 mod 8 (every PU shape), block inside the
 * picture (else cost 0xFFFFFFFF). */
typedef struct {
  int32_t x, y, width, height;
  int16_t mv0[2], mv1[2];
} kvz_hip_bipred_cand;
KVZ_HIP_API int kvz_hip_bipred_cost_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, int pic_w, int pic_h,
                                          const kvz_hip_pixel *ref0, uint32_t ref0_stride,
                                          const kvz_hip_pixel *ref1, uint32_t ref1_stride, int ref_w, int ref_h,
                                          const kvz_hip_bipred_cand *cands, size_t count, uint32_t *costs, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* (2) batched entries -- intra group (strategies/strategies-intra.h)  */
/*     SURVEY.md section 8(f) row 2                                    */
/* ------------------------------------------------------------------ */
/* kvz_intra_ref (intra.h:35-38): reference pixels of one PU, entry 0 of both
 * arrays = the top-left corner, entries 1..2N the left / top neighbours. */
typedef struct {
  kvz_hip_pixel left[2 * 32 + 1];
  kvz_hip_pixel top[2 * 32 + 1];
} kvz_hip_intra_ref;

#define KVZ_HIP_INTRA_LUMA 1             /* color == COLOR_Y */
#define KVZ_HIP_INTRA_FILTER_BOUNDARY 2  /* kvz_intra_predict's filter_boundary */
#define KVZ_HIP_INTRA_RAW 4              /* the bare strategies kvz_angular_pred / kvz_intra_pred_planar
                                            (intra-generic.c:37-189) on the given references: no smoothing
                                            decision, no DC / boundary filters */

/* Luma position of an intra PU in its (tile's) picture: the luma_px argument of
 * kvz_intra_build_reference (intra.h:94-100). */
typedef struct { int32_t x, y; } kvz_hip_intra_pos;

/* kvz_intra_build_reference (intra.c:334-588) for every listed PU, gathered on
 * the device from `rec`, the plane of `color` (0 Y, 1 U, 2 V; stride in pixels
 * of that plane) of the reconstruction BEFORE deblocking -- what lcu->rec,
 * lcu->top_ref and lcu->left_ref are views of (init_lcu_t,
 * search.c:761-835).  pic_width / pic_height are pic_px, the luma size of
 * the (tile) picture (multiples of 8, as the encoder pads them).  Which neighbours count as coded follows from the PU's
 * place in the coding order of its LCU exactly as num_ref_pixels_top / _left
 * (intra.c:35-70) say, so pixels of CUs that come later are never read and may
 * hold anything.  Entries 0..2N of both arrays are the reference's, the rest
 * is zero.  A position outside the picture or off the 4-pixel grid yields an
 * all-zero record.  refs feeds kvz_hip_intra_predict_batch / _rough_batch on
 * the same stream without a host round trip. */
KVZ_HIP_API int kvz_hip_intra_build_reference_batch(int log2_width, int color, const kvz_hip_pixel *rec, int stride,
                                                    int pic_width, int pic_height, const kvz_hip_intra_pos *pus, size_t count,
                                                    kvz_hip_intra_ref *refs, kvz_hip_stream s);

/* kvz_intra_predict (intra.c:281-331) for every PU x every listed mode:
 * reference smoothing (intra.c:164-192) chosen per mode and size, planar,
 * DC with its edge filter (intra.c:217-278), angular modes 2..34 with the
 * boundary post-process of modes 10 / 26 (intra.c:195-208).  modes is a HOST
 * array of num_modes (1..35) mode numbers; prediction (i, k) is written as
 * N*N contiguous pixels at dst + (i * num_modes + k) * N * N. */
KVZ_HIP_API int kvz_hip_intra_predict_batch(int log2_width, int flags, const kvz_hip_intra_ref *refs, size_t count,
                                            const int8_t *modes, int num_modes, kvz_hip_pixel *dst, kvz_hip_stream s);

/* The cost loop of search_intra_rough (search_intra.c:404-520) for all 35
 * modes at once: satd_costs[35*i + m] = satd_NxN(kvz_intra_predict(mode m), orig_i)
 * (luma), orig = count contiguous N x N blocks.  sad_costs (NULL, or same shape)
 * receives sad_NxN for the 4x4 transform-skip test of get_cost
 * (search_intra.c:99-126).  The host walks the table in the reference's order
 * and adds the mode-bit cost, so decisions are identical; predictions never
 * leave the CU. */
KVZ_HIP_API int kvz_hip_intra_rough_batch(int log2_width, int flags, const kvz_hip_intra_ref *refs,
                                          const kvz_hip_pixel *orig, size_t count,
                                          uint32_t *satd_costs, uint32_t *sad_costs, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* (2) batched entries -- SAO group (strategies/strategies-sao.h)      */
/*     SURVEY.md section 8(f) row 4                                    */
/* ------------------------------------------------------------------ */
/* Statistics / distortion entries: `count` contiguous block_width x
 * block_height blocks (stride = block_width, at most 64 x 64), the way
 * sao.c blits an LCU plane before calling the strategies. */

/* calc_sao_edge_dir (sao-generic.c:80-109) for the four edge classes at once:
 * cat_sum_cnt[((i * 4 + eo_class) * 2 + {0 sum, 1 count}) * 5 + category]. */
KVZ_HIP_API int kvz_hip_sao_edge_stats_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height,
                                             size_t count, int32_t *cat_sum_cnt, kvz_hip_stream s);
/* sao_edge_ddistortion (sao-generic.c:46-77) for the four classes:
 * ddistortion[i * 4 + eo_class] with offsets[(i * 4 + eo_class) * 5 + category]. */
KVZ_HIP_API int kvz_hip_sao_edge_ddistortion_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height,
                                                   size_t count, const int32_t *offsets, int32_t *ddistortion, kvz_hip_stream s);
/* calc_sao_bands (sao.c:247-261): sao_bands[(i * 2 + {0 sum, 1 count}) * 32 + band] */
KVZ_HIP_API int kvz_hip_sao_band_stats_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height,
                                             size_t count, int32_t *sao_bands, kvz_hip_stream s);
/* sao_band_ddistortion (sao-generic.c:157-183): ddistortion[i] for band_pos[i], sao_bands[i * 4 + k] */
KVZ_HIP_API int kvz_hip_sao_band_ddistortion_batch(const kvz_hip_pixel *orig, const kvz_hip_pixel *rec, int block_width, int block_height,
                                                   size_t count, const int32_t *band_pos, const int32_t *sao_bands,
                                                   int32_t *ddistortion, kvz_hip_stream s);

/* the fields of sao_info_t (sao.h:42-50) the reconstruction reads */
typedef struct {
  int32_t type;                 /* SAO_TYPE_NONE 0 (copy), SAO_TYPE_BAND 1, SAO_TYPE_EDGE 2 */
  int32_t eo_class;
  int32_t band_position[2];     /* [0] Y / U, [1] V */
  int32_t offsets[10];          /* [0..4] Y / U, [5..9] V */
} kvz_hip_sao_info;
typedef struct { int32_t x, y, width, height, sao_index; } kvz_hip_sao_block;
/* sao_reconstruct_color (sao-generic.c:112-154, with kvz_calc_sao_offset_array,
 * sao.c:164-180) for `count` blocks of one plane: new_rec <- filter(rec) inside
 * each block.  Edge blocks read one pixel beyond the block in the directions of
 * their class, so the caller trims them at the picture border like
 * kvz_sao_reconstruct (sao.c:296-318); a descriptor whose reads would leave the
 * plane is skipped.  color 0 Y, 1 U, 2 V. */
KVZ_HIP_API int kvz_hip_sao_reconstruct_color_batch(const kvz_hip_pixel *rec, uint32_t stride, int plane_w, int plane_h,
                                                    kvz_hip_pixel *new_rec, uint32_t new_stride,
                                                    const kvz_hip_sao_block *blocks, size_t count,
                                                    const kvz_hip_sao_info *infos, int n_infos, int color, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* deblocking of a reconstructed frame                                 */
/*   reference: kvz_filter_deblock_lcu (filter.c:770-779) called for   */
/*   every LCU (encoderstate.c:579-616), with filter.c:83-768 below it */
/*   SURVEY.md section 8(f) row 4                                      */
/* ------------------------------------------------------------------ */
/* cus: kvz_hip_cu_info records (defined with the candidate derivation above), one per 4x4 SCU, row-major,
 * ceil(width / 4) records per row. */
typedef struct {
  int32_t beta_offset_div2;     /* cfg.deblock_beta */
  int32_t tc_offset_div2;       /* cfg.deblock_tc */
  int32_t qp;                   /* state->qp, used when per_cu_qp == 0 (get_qp_y_pred, filter.c:263-282) */
  int32_t frame_qp;             /* state->frame->QP */
  int32_t per_cu_qp;            /* encoder_control->max_qp_delta_depth >= 0 */
  int32_t slice_is_b;           /* state->frame->slicetype == KVZ_SLICE_B */
  int32_t chroma;               /* 0: 4:0:0 (rec_u / rec_v unused), 1: 4:2:0 */
  int32_t reserved;
  uint8_t ref_LX[2][16];        /* state->frame->ref_LX (encoderstate.h:100) */
} kvz_hip_deblock_params;       /* 64 bytes */
/* Filters the planes in place: every vertical edge of the frame, then every
 * horizontal edge (two launches) -- the order the reference's LCU walk with its
 * deferred rightmost 4 pixels (filter.c:711-779) implements.  width / height
 * multiples of 8, planes / strides / cus 4-byte aligned; bitdepth 8, lossless
 * and PCM blocks are not handled (kvz_filter_deblock_lcu asserts !lossless). */
KVZ_HIP_API int kvz_hip_deblock_frame(kvz_hip_pixel *rec_y, uint32_t stride_y, kvz_hip_pixel *rec_u, kvz_hip_pixel *rec_v,
                                      uint32_t stride_c, int width, int height, const kvz_hip_cu_info *cus,
                                      const kvz_hip_deblock_params *params, kvz_hip_stream s);

/* ------------------------------------------------------------------ */
/* (1) strategy registration -- the drop-in boundary                   */
/* ------------------------------------------------------------------ */
/* kvz_strategyselector_register (strategyselector.h:87, strategyselector.c:216-256) */
typedef int (*kvz_hip_register_fn)(void *opaque, const char *type, const char *strategy_name, int priority, void *fptr);
/* By default the hooks call the process's own kvz_strategyselector_register
 * (resolved at load time from the host encoder).  A host that links the
 * selector with hidden visibility passes it explicitly. */
KVZ_HIP_API void kvz_hip_set_registrar(kvz_hip_register_fn fn);
/* Number of per-call strategy invocations that launched GPU work since load (all threads):
 * lets a host or test confirm the "hip" pointers are the ones its encoder is calling. */
KVZ_HIP_API unsigned long long kvz_hip_dropin_calls(void);

/* Accessors for the opaque encoder_state_t the quant strategies receive
 * (encoderstate.h; quant-generic.c:40-50).  Supplied by the few lines of glue
 * compiled inside Kvazaar; without them the quant group
 * registers only coeff_abs_sum. */
typedef struct {
  int (*qp)(const void *state);
  int (*slice_is_intra)(const void *state);
  int (*signhide_enable)(const void *state);
  int (*scaling_list_enable)(const void *state);
  const int32_t *(*quant_coeff)(const void *state, int log2_tr_size, int list_type, int qp_rem);
  const int32_t *(*dequant_coeff)(const void *state, int log2_tr_size, int list_type, int qp_rem);
  int (*rdoq_enable)(const void *state);
  /* cu_info_t fields used by quantize_residual (quant-generic.c:197-225) */
  int (*cu_is_intra)(const void *cur_cu);
  /* optional (may be NULL): planes of hi_prec_buf_t (image.h:38-46) and of lcu_t.rec (cu.h:287-325)
   * for inter_recon_bipred; without them that one function stays on the CPU strategy */
  const int16_t *(*hi_prec_y)(const void *hi_prec_buf);
  const int16_t *(*hi_prec_u)(const void *hi_prec_buf);
  const int16_t *(*hi_prec_v)(const void *hi_prec_buf);
  kvz_hip_pixel *(*lcu_rec_y)(void *lcu);
  kvz_hip_pixel *(*lcu_rec_u)(void *lcu);
  kvz_hip_pixel *(*lcu_rec_v)(void *lcu);
  /* What quantize_residual needs when cfg.rdoq_enable (quant-generic.c:214-221).  kvz_rdoq (rdo.c:548) is the
   * encoder's CABAC-context dependent quantiser -- control plane, not part of this library; like the avx2 strategy,
   * the hip quantize_residual calls the host's own function between the transform (or transform skip) and the
   * dequantisation.  quantize_residual is registered only when rdoq_enable and these four are all supplied, or when
   * rdoq_enable itself is NULL (= the host vouches that RDOQ is never on); likewise quant / dequant /
   * quantize_residual need quant_coeff and dequant_coeff whenever scaling_list_enable is supplied.  A group function
   * that is not registered stays on the host's next-best strategy -- nothing diverges or aborts at run time. */
  int (*rdoq_skip)(const void *state);                 /* cfg.rdoq_skip */
  int (*cu_rdoq_tr_depth)(const void *cur_cu);         /* tr_depth - depth + (part_size == SIZE_NxN) */
  int (*cu_type)(const void *cur_cu);                  /* cur_cu->type, handed to kvz_rdoq as block_type */
  void (*rdoq)(void *state, kvz_hip_coeff *coef, kvz_hip_coeff *dest_coeff, int32_t width, int32_t height,
               int8_t type, int8_t scan_mode, int8_t block_type, int8_t tr_depth);
} kvz_hip_state_accessors;
KVZ_HIP_API void kvz_hip_set_state_accessors(const kvz_hip_state_accessors *acc);

#define KVZ_HIP_STRATEGY_NAME "hip"
#define KVZ_HIP_STRATEGY_PRIORITY 50     /* avx2 = 40 (picture-avx2.c:1232) */

/* Same shape as kvz_strategy_register_picture_avx2 (picture-avx2.c:1224):
 * registers every function of the group under the type strings of
 * STRATEGIES_<GROUP>_EXPORTS, name "hip", priority 50; only for bitdepth 8.
 * Returns 1 on success, 0 on failure. */
KVZ_HIP_API int kvz_strategy_register_picture_hip(void *opaque, uint8_t bitdepth);
KVZ_HIP_API int kvz_strategy_register_dct_hip(void *opaque, uint8_t bitdepth);
KVZ_HIP_API int kvz_strategy_register_quant_hip(void *opaque, uint8_t bitdepth);
KVZ_HIP_API int kvz_strategy_register_ipol_hip(void *opaque, uint8_t bitdepth);
KVZ_HIP_API int kvz_strategy_register_intra_hip(void *opaque, uint8_t bitdepth);   /* strategies-intra.h:52-55 */
KVZ_HIP_API int kvz_strategy_register_sao_hip(void *opaque, uint8_t bitdepth);     /* strategies-sao.h:64-69 */

#ifdef __cplusplus
}
#endif
#endif /* KVZ_HIP_H_ */
