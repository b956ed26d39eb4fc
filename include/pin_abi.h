/* pin_abi.h -- C ABI of libpinhip.so: the MI355X (gfx950) implementation of PIN-SLAM's
 * per-frame neural-point SDF hot path.
 *
 * The reference (PRBonn/PIN_SLAM) has no FFI: its boundary for this path is the Python
 * class surface model/neural_points.py::NeuralPoints, model/decoder.py::Decoder,
 * utils/mapper.py::Mapper.mapping and utils/tracker.py::Tracker.tracking.  Every entry
 * point below cites the reference function (file:line, relative to the reference tree)
 * whose tensor-op chain it replaces.  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes, no torch types.  All array pointers are DEVICE pointers
 *    unless the name ends in _host.  Memory is owned by the caller; the library never
 *    allocates device memory; scratch comes from caller-provided workspaces.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream) and returns 0 on success, <0 on error; pin_last_error() gives text.
 *  - indices are int32 on the device (the reference uses int64 tensors); -1 = invalid.
 *  - float arithmetic is IEEE fp32 with the reference's evaluation order wherever an
 *    integer result depends on it (voxel coordinates, neighbour distances).
 */
#ifndef PIN_ABI_H
#define PIN_ABI_H

#include <stdint.h>

/* The sample pool of Mapper (utils/mapper.py:76-97: coord_pool, global_coord_pool,
 * sdf_label_pool, weight_pool, time_pool, color_pool), capacity-managed by the caller; the
 * pointers passed to pin_sample_rays are already advanced to the append position. */
typedef struct pin_pool_arrays {
    float* coord;         /* [n][3] sensor-frame sample positions */
    float* global_coord;  /* [n][3] transform_torch(coord, pose of the frame) */
    float* sdf_label;     /* [n]    projective signed distance (behind +, in front -) */
    float* weight;        /* [n]    sample weight, negative = free-space sample */
    int32_t* ts;          /* [n]    frame id */
    float* color;         /* [n][color_channels] or NULL */
    int32_t color_channels;
    int32_t reserved;
    int32_t* sem_label;   /* [n] semantic label of the sample (sem_label_pool, utils/mapper.py:92, 280-283) or NULL */
} pin_pool_arrays;

/* DataSampler.sample configuration (utils/data_sampler.py:26-36, config.py:124-129,169-171).
 * Values are the reference's python floats (double); the library casts them to float32 exactly
 * where torch does. */
typedef struct pin_sample_params {
    int32_t surface_n, front_n, behind_n;   /* surface_sample_n, free_front_n, free_behind_n */
    int32_t dist_weight_on, behind_dropoff_on;
    int32_t frame_id;                       /* written to the time pool */
    double surface_range;                   /* surface_sample_range_m */
    double free_begin_ratio, free_end_dist; /* free_sample_begin_ratio, free_sample_end_dist_m */
    double dist_weight_scale, max_range;
    double pose[12];                        /* sensor pose rows [R|t] (cur_pose_torch[:3,:]) */
    const int32_t* sem_labels;              /* DEVICE [n] or NULL: per-point semantic labels of the scan (frame_label_torch);
                                             * the measured point and its close-to-surface samples inherit them, free-space
                                             * samples get label 0 (utils/data_sampler.py:59-62, 83-84, 105-106, 184-194) */
} pin_sample_params;

#ifdef __cplusplus
extern "C" {
#endif

#define PIN_ABI_VERSION 18
#define PIN_FEATURE_DIM 8          /* config.feature_dim (utils/config.py:103) */
#define PIN_MLP_IN (PIN_FEATURE_DIM + 3)
#define PIN_MAX_K 8                /* query_nn_k: 6 default, 8 in the benchmark configs */
#define PIN_NONLOCAL (-2)          /* global2local value of a non-local point, see below */
#define PIN_NBR_QUIRK_BIT 0x40000000
#define PIN_ADAM_ROW_EXCLUDED (-2147483647 - 1) /* pending word of a feature row the lazy optimiser must leave alone */

/* ---- voxel-hash search state: NeuralPoints fields used by
 * radius_neighborhood_search (model/neural_points.py:950-1009) ------------------- */
typedef struct pin_search_params {
    const int32_t* table;        /* [buffer_size] slot -> global point index, -1 = empty
                                    (buffer_pt_index, neural_points.py:88; int64 there) */
    const float*   pos4;         /* [n_points][4] x, y, z, bit pattern of point_ts_create:
                                    one 16-byte gather per occupied candidate cell
                                    (neural_points, point_ts_create: neural_points.py:92,112) */
    const int32_t* cand_off;     /* [Kc] (dx*p0 + dy*p1 + dz*p2) mod buffer_size for the
                                    candidate offsets neighbor_dx (neural_points.py:919-932),
                                    from pin_candidate_offsets() */
    const float*   travel_dist;  /* [n_ts] accumulated travel distance per frame, or NULL =
                                    no travel-distance window filter (neural_points.py:982-988) */
    const int32_t* global2local; /* [n_points+1] or NULL = query the global map.  Values:
                                    local index, -1 (padding entry), PIN_NONLOCAL for points
                                    outside the local map.  The reference maps those to local
                                    index 1 (torch.full_like(bool, -1).long(), neural_points.py:498);
                                    the kernels reproduce that and flag the neighbour. */
    int64_t buffer_size;         /* config.buffer_size, < 2^31 */
    int32_t n_points;
    int32_t n_cand;              /* Kc = neighbor_K */
    int32_t cur_ts;              /* NeuralPoints.cur_ts */
    float   diff_travel_dist_local; /* local_map_radius * local_map_travel_dist_ratio */
    float   resolution;          /* voxel_size_m */
    float   max_valid_dist2;     /* 3*((num_nei_cells+1)*resolution)^2 (neural_points.py:947) */
} pin_search_params;

/* ---- brick cache: a cell-coherent, per-frame view of the SAME lookups (csrc/brick.hip) ----
 * Built by pin_brick_build from a pin_search_params (its travel_dist / global2local settings
 * are baked in); pin_knn_query_bricks must be called with the same search params and returns
 * results identical to pin_knn_query.  All buffers are caller-owned. */
typedef struct pin_brick_cache {
    uint64_t* dir_keys;      /* [dir_mask+1] open-addressing directory: packed brick coordinate */
    int32_t*  dir_vals;      /* [dir_mask+1] brick id */
    uint64_t* brick_keys;    /* [max_bricks] */
    uint64_t* brick_mask;    /* [max_bricks] occupancy of the 4x4x4 cells, bit = (x&3)<<4|(y&3)<<2|(z&3) */
    int32_t*  brick_base;    /* [max_bricks] first entry of the brick */
    float*    entries;       /* [max_entries + 1][4] x, y, z, neighbour-index bits (as in the kNN record); row max_entries is
                                a sentinel (+inf coordinates) written by pin_brick_build: what an empty cell reads */
    const int32_t* cand_dx;  /* [n_cand][3] candidate cell offsets (neighbor_dx, neural_points.py:919-932) */
    uint32_t dir_mask;       /* directory size - 1 (size is a power of two) */
    int32_t max_bricks;
    int32_t max_entries;
    int32_t n_dilate;        /* = num_nei_cells (<= 2): bricks cover every cell within n of a local point */
    uint64_t* dir_pack;      /* [dir_mask+1][4] what a query reads: key, mask, base (low 32 bits; -1 = not cached), pad --
                                one 32-byte slot per probe instead of the key -> id -> mask/base chain */
    void* build_ws;          /* scratch of pin_brick_build (pin_brick_build_workspace_bytes(n_points, max_bricks)); with it the
                                cache is built from the POINTS (one table probe per point instead of one per cell of every brick;
                                same directory contents, masks and entries) -- NULL: the cell-driven build */
    int64_t build_ws_bytes;
    int32_t build_grid;      /* > 0: at most this many 256-thread blocks per launch of pin_brick_build, each walking its share of
                                the work (the build is bound by random table probes: a few waves per compute unit keep the memory
                                system busy and leave the dispatcher's slots to the launches of a stream that runs beside it);
                                0: one block per unit of work */
    int32_t pad_;
} pin_brick_cache;

/* ---- the implicit field: feature tables + decoder (NeuralPoints.query_feature
 * neural_points.py:530-746, Decoder.mlp/sdf model/decoder.py:61-85) -------------- */
typedef struct pin_field {
    const float* feats;      /* [M+1][8] geo features of the index space that was searched
                                (local_geo_features or geo_features; last row = padding) */
    const float* certainty;  /* [M] point certainties or NULL */
    const float* orient;     /* [M][4] quaternions, non-NULL only after PGO (after_pgo,
                                neural_points.py:645-648) */
    const float* pos;        /* [M][3] positions of that index space (used only for flagged
                                neighbours and after PGO) */
    const float* dec;        /* flat decoder parameters in state_dict order: layers.i.weight
                                [H][in] row-major, layers.i.bias [H], lout.weight [1][H], lout.bias */
    int32_t k;               /* query_nn_k, <= PIN_MAX_K */
    int32_t hidden;          /* 32 or 64 */
    int32_t levels;          /* hidden layers, 1..4 */
    int32_t weighted_first;  /* config.weighted_first (utils/config.py:93) */
    float   sdf_scale;       /* logistic_gaussian_ratio * sigma_sigmoid_m (decoder.py:54-56) */
    int32_t out_dim;         /* decoder heads: 1 (sdf; 0 means 1) or 3 (colour, Decoder.regress_color) */
    int32_t dec_image_bytes; /* size of dec_image, 0 if none */
    const void* dec_image;   /* optional: the decoder as the Gauss-Newton tile kernel lays it out in LDS (split-fp16
                                pieces, both directions), written by pin_stage_decoder for THIS dec / hidden / levels / out_dim.
                                The kernel then copies it instead of re-splitting `dec` in every block of every launch;
                                restage whenever the decoder parameters change.  NULL = stage from `dec`. */
} pin_field;

/* pin_field.dec_image: size for a decoder shape (0 if that shape has no staged form), and the staging launch. */
int64_t pin_decoder_image_bytes(int32_t hidden, int32_t levels);
int pin_stage_decoder(const pin_field* f, void* image_out, int64_t image_bytes, void* stream);

/* ---- Gauss-Newton registration (Tracker.registration_step + implicit_reg,
 * utils/tracker.py:409-524, 615-695) --------------------------------------------- */
typedef struct pin_gn_params {
    int32_t valid_nn_k;      /* track_mask_query_nn_k */
    float min_grad_norm, max_grad_norm;   /* reg_min/max_grad_norm */
    float max_sdf_std;       /* surface_sample_range_m * max_sdf_std_ratio */
    float gm_dist;           /* reg_GM_dist_m, <=0 disables */
    float gm_grad;           /* reg_GM_grad, <=0 disables */
    int32_t dist_div_grad_norm; /* reg_dist_div_grad_norm (utils/tracker.py:452-456): the residual uses sdf / |grad|
                                (the Jacobian rows and the Geman-McClure gradient weight keep the plain values) */
} pin_gn_params;
/* optional colour term of the registration (utils/tracker.py:493-542, 699-744) */
typedef struct pin_color_term {
    const pin_field* field;  /* colour feature table + colour decoder (out_dim 3), same k / hidden /
                                weighting mode as the sdf field */
    const float* colors;     /* [n][3] measured colours of the source points (device) */
    int32_t mode;            /* 0 off; 1 consistency weight exp(-|I_meas - I_pred|) (consist_wieght_on);
                                2 photometric term (photometric_loss_on) */
    float photo_weight;      /* photometric_loss_weight */
} pin_color_term;
#define PIN_GN_NSUMS 32
#define PIN_GN_REPLICAS 16
/* Device-resident state of one Tracker.tracking call (utils/tracker.py:114-184), doubles:
 * [0..15] current pose T (row-major 4x4)   [16] last_sdf_residual_cm   [17] sdf_residual_cm
 * [18] valid point count   [19] valid_flag   [20] converged   [21] done (loop has ended)
 * [22] iterations run      [23] weighted mse (cov scale)     [24..59] un-damped J^T W J (6x6)
 * [60] source point count */
#define PIN_GN_STATE_DOUBLES 80
#define PIN_GN_STATE_LAST_RES 16
#define PIN_GN_STATE_RES 17
#define PIN_GN_STATE_CNT 18
#define PIN_GN_STATE_VALID 19
#define PIN_GN_STATE_CONVERGED 20
#define PIN_GN_STATE_DONE 21
#define PIN_GN_STATE_ITERS 22
#define PIN_GN_STATE_MSE 23
#define PIN_GN_STATE_NRAW 24
#define PIN_GN_STATE_NSRC 60
#define PIN_GN_STATE_STATUS 61   /* the library's sticky status flags (pin_status) as the solve kernel last saw them */
#define PIN_GN_STATE_TICKET 62   /* 64-bit block counter of the tile kernels that finish the iteration themselves (pin_gn_accumulate_solve); 0 between launches */
#define PIN_GN_STATE_LP 64       /* pin_gn_loop_init: the loop parameters as 8 doubles, in the order of pin_gn_loop_params */
#define PIN_GN_STATE_STATUS_PTR 72  /* pin_gn_loop_init: bit pattern of the device address of the library's status word */

typedef struct pin_gn_loop_params {      /* Tracker.tracking constants (tracker.py:77-101) */
    double lm_lambda;                    /* reg_lm_lambda */
    double term_thre_deg, term_thre_m;   /* reg_term_thre_deg / _m */
    double min_valid_ratio;              /* 0.2, 0.15 for loop registration */
    double max_increment_ratio;          /* 1.1 */
    int32_t min_valid_points;            /* 30 */
    int32_t iter_n;                      /* reg_iter_n */
    int32_t early_exit;                  /* 0: ignore the convergence test (benchmark worst case) */
} pin_gn_loop_params;

/* sums layout (double[PIN_GN_NSUMS]): [0..20] upper triangle of sum w J J^T (row-major,
 * J = [p x g, g]); [21..26] sum w J r; [27] sum w; [28] sum |r|; [29] valid count;
 * [30] sum w r^2; [31] sum |I_pred - I_meas| (photometric mode).  w is the un-normalised robust weight; the host applies
 * the reference's w /= 2*mean(w) (tracker.py:524) as one scalar. */

/* ---- map training (Mapper.mapping loop body, utils/mapper.py:645-818) ------------ */
typedef struct pin_train_params {
    int32_t n_main;          /* batch size bs (config.bs) */
    int32_t n_eik;           /* ceil(bs / gradient_decimation) Eikonal samples, 0 = Eikonal off */
    int32_t loss_weight_on;  /* config.loss_weight_on */
    float sigma;             /* BCE sigmoid scale = Mapper.sdf_scale (mapper.py:66) */
    float weight_e;          /* config.weight_e */
    float eik_eps;           /* voxel_size_m * num_grad_step_ratio (mapper.py:685) */
    float inv_n_main;        /* 1 / GLOBAL batch size  (the loss means; sharded batches pass */
    float inv_n_eik;         /* 1 / GLOBAL Eikonal count  the global counts)      */
    int32_t eik_analytic;    /* numerical_grad_on False (config.py:437-439, run_livox.yaml:27): the Eikonal term on the
                              * autograd gradient of EVERY main sample (mapper.py:642-643, 677-678, 760-782) and its second
                              * derivative; n_eik = 0 (no probes), inv_n_eik = 1 / GLOBAL batch size, the workspace sized
                              * for 2 * n_main queries.  Built for weighted_first = 0 with a one-layer decoder. */
    int32_t dec_image_current;/* pin_field.dec_image holds THIS decoder's current parameters (staged by pin_stage_decoder
                              * and kept current by pin_adam_dense.image): skip the staging launch */
    int32_t defer_weight_grad;/* pin_train_step stops after the tile kernel (feature gradients done, operand stream in the
                              * workspace); pin_train_weight_grad finishes the step.  Fused tile paths only. */
    int32_t defer_dec_reduce; /* leave the decoder's weight gradient of this step as the weight-gradient launch wrote it -- slot copies in
                              * the workspace -- instead of adding it into dec_grad with a launch of its own: the optimiser's decoder
                              * step takes it from there (pin_adam_dense.grad_partial; pin_train_deferred_partial says where it is).
                              * The loss sums of the step are not formed (loss_out keeps its value).  Fused tile paths only. */
} pin_train_params;

/* ---- map maintenance (NeuralPoints.update / reset_local_map / assign_local_to_global,
 * model/neural_points.py:311-526) ------------------------------------------------- */
typedef struct pin_map_arrays {      /* the global map, capacity-managed SoA (neural_points.py:88-118) */
    int32_t* table;        /* [buffer_size] */
    float*   pos;          /* [cap][3]  neural_points */
    float*   pos4;         /* [cap][4]  packed search mirror (xyz, ts_create bits) */
    float*   orient;       /* [cap][4]  point_orientations (w,x,y,z) */
    float*   geo;          /* [cap+1][8] geo_features, row n_points = padding */
    float*   color;        /* [cap+1][8] color_features or NULL */
    int32_t* ts_create;    /* [cap] */
    int32_t* ts_update;    /* [cap] */
    float*   certainty;    /* [cap] */
} pin_map_arrays;

typedef struct pin_local_arrays {    /* the local map (neural_points.py:120-136) */
    float*   pos;          /* [M][3] */
    float*   orient;       /* [M][4] */
    float*   geo;          /* [M+1][8] local_geo_features (+ padding row) */
    float*   color;        /* [M+1][8] or NULL */
    float*   certainty;    /* [M] */
    int32_t* ts_update;    /* [M] */
    int32_t* global2local; /* [n_points+1], PIN_NONLOCAL for non-local points, -1 padding */
} pin_local_arrays;

typedef struct pin_update_params {
    const float* travel_dist;  /* [n_ts] or NULL (temporal_local_map_on == False) */
    int64_t buffer_size;
    int32_t n_points;          /* points in the map before the call */
    int32_t capacity;          /* rows allocated in pin_map_arrays */
    int32_t n_max;             /* upper bound of *n_sel (size of sel) */
    int32_t cur_ts;
    int32_t all_new;           /* empty map or cur_ts == reboot_ts: every sample becomes a point (:341,357) */
    float resolution;
    float dist2_thre;          /* 3 * resolution^2 (:345) */
    float diff_travel_dist_local;
} pin_update_params;

typedef struct pin_local_params {
    const float* travel_dist;  /* use_travel_dist: accumulated travel distance per frame (:451-455) */
    int32_t n_points;
    int32_t cur_ts;
    int32_t reboot_ts;         /* >= 0 only when reboot_map (:465-466), else -1 */
    float diff_travel_dist_local;
    int32_t time_mode;         /* 0 = no time mask (temporal_local_map_on False), 1 = travel distance window
                                  |travel[cur_ts] - travel[ts]| < diff_travel_dist_local (:451-455), 2 = frame window
                                  |cur_ts - ts| < diff_ts_local (use_travel_dist False, :456-458) */
    int32_t diff_ts_local;
    int32_t use_mid_ts;        /* config.use_mid_ts: ts = ((ts_create + ts_update) / 2).int() (:443-447) */
    int32_t sensor_f64;        /* the caller's sensor_position is float64: `neural_points - sensor_position` promotes,
                                  the radius test runs in float64 (:476-479); 0 = float32 arithmetic */
    double sensor[3];
    double radius2;            /* local_map_radius ** 2 (rounded to float32 by the library when sensor_f64 == 0) */
} pin_local_params;

/* recreate_hash / prune_map (model/neural_points.py:748-789, 819-908) */
typedef struct pin_rehash_params {
    int64_t buffer_size;
    int32_t n_points;
    int32_t cur_ts;
    int32_t with_ts;           /* 1: voxel winner = smallest |ts - cur_ts| (:843-851); 0: largest certainty (:853-858) */
    int32_t use_mid_ts;
    float resolution;
} pin_rehash_params;

typedef struct pin_prune_params {
    const float* travel_dist;  /* used unless global_prune */
    int32_t n_points;
    int32_t cur_ts;
    int32_t global_prune;
    float certainty_thre;
    float diff_travel_dist_local;
} pin_prune_params;

typedef struct pin_train_color_params {   /* colour term of Mapper.mapping (mapper.py:668-675, 804-812) */
    int32_t n_main;          /* batch size */
    int32_t loss_weight_on;  /* config.loss_weight_on */
    float surface_range;     /* surface_sample_range_m: samples with |sdf_label| below it carry colour */
    float weight_i;          /* config.weight_i */
    int32_t dec_image_current;/* as in pin_train_params, for the colour field's dec_image */
    int32_t n_main_global;   /* with surface_count: size of the GLOBAL batch this call is a shard of (0 = n_main); it bounds the
                              * loss gradients from below (every global sample on the surface), which picks the power-of-two
                              * scale of the fp16 backward sweep */
    const int32_t* surface_count; /* DEVICE, may be NULL: the number of surface samples of the GLOBAL batch when this call
                              * evaluates a shard of it (the colour loss is a mean over them, utils/loss.py:31-42);
                              * NULL = count the samples of this call */
} pin_train_color_params;

/* ---- semantic head (config.semantic_on, config/lidar_slam/run_demo_sem.yaml) ------------------------------------------------
 * A decoder with heads = sem_class_count + 1 outputs over the SAME interpolated geometry feature as the SDF decoder
 * (pin_field.feats = geometry features, pin_field.dec = the semantic decoder's flat parameters, state_dict order), read through
 * a log-softmax: Decoder.sem_label_prob (model/decoder.py:100-103). */
typedef struct pin_sem_params {
    int32_t n_main;          /* batch size */
    int32_t heads;           /* sem_class_count + 1, 2..32 */
    float weight_s;          /* config.weight_s */
    int32_t reserved;
    const int32_t* labels;   /* DEVICE [n_main] semantic label of every batch sample (Mapper.get_batch's sem_label) */
    const uint8_t* selected; /* DEVICE [n_main] 1 = the sample is in the loss (pin_sem_select) */
    const int32_t* count;    /* DEVICE: number of selected samples, the mean's denominator (pin_sem_select) */
} pin_sem_params;
int64_t pin_sem_workspace_bytes(int32_t n_queries, int32_t hidden, int32_t levels, int32_t expand);
/* utils/mapper.py:786-799: label_mask = label > 0 (>= 0 with freespace_label_on); of the masked samples, in order, every
 * decimation-th (sem_pred[label_mask][::sem_label_decimation]).  selected_out [n] in {0, 1}, count_out [1] = number selected. */
int pin_sem_select(const int32_t* labels, int32_t n, int32_t freespace_label_on, int32_t decimation, uint8_t* selected_out,
                   int32_t* count_out, void* stream);
/* Mapper.get_batch's sem_label gather (utils/mapper.py:490-491) with the index arrays of pin_gather_batches_drawn; out [n_batches][n]. */
int pin_gather_labels_drawn(const int32_t* pool_sem, const int64_t* index_history, int32_t n_history, const int64_t* index_new_batch,
                            const int64_t* new_idx, int32_t n, int32_t n_batches, int64_t hist_stride, int64_t new_stride,
                            int32_t* out, void* stream);
/* The semantic term of one Mapper.mapping iteration (utils/mapper.py:664-667, 782-800) on the queries / kNN records of
 * pin_train_step's main samples: sem_pred = log_softmax(mlp(feature)) (per neighbour and weighted when !weighted_first), NLL
 * mean over the selected samples x weight_s; gradients ADD into feat_grad (the geometry feature table's, shared with
 * pin_train_step) and dec_grad (the semantic decoder's, may be NULL = frozen); loss_out[0] += sum over the selected samples of
 * -sem_pred[label] (divide by *count for the reference's mean). */
int pin_train_sem_step(const pin_field* f, const pin_sem_params* sp, const float* query, const float* nbr, const int32_t* nn_count,
                       float* feat_grad, float* dec_grad, double* loss_out, void* workspace, int64_t workspace_bytes, void* stream);
/* Tracker.query_source_points(query_sem) / Mesher.query_points(query_sem) (utils/tracker.py:336-341, utils/mesher.py:137-145):
 * label_out [n] = argmax of the (weighted) log-probabilities, logprob_out [n][heads] the log-probabilities; either may be NULL. */
int pin_sem_query(const pin_field* f, const float* query, const float* nbr, const int32_t* nn_count, int32_t n, int32_t heads,
                  int32_t* label_out, float* logprob_out, void* stream);
/* Decoder.sem_label_prob on given decoder inputs feat_in [n][11] -> out [n][heads] (raw != 0: Decoder.mlp's plain outputs). */
int pin_decoder_sem(const pin_field* f, const float* feat_in, int32_t n, int32_t heads, int32_t raw, float* out, void* stream);

/* ---- library ------------------------------------------------------------------- */
int         pin_version(void);
/* Load every code object of the library now (the HIP runtime otherwise does it inside the first launch of each translation
 * unit's kernels -- a 100 ms frame the first time a stage of the SLAM loop runs).  Needs a current device; no kernel runs. */
int         pin_warmup(void);
const char* pin_last_error(void);
/* Sticky status flags the kernels raise on the device (one word per device).  pin_status copies the word to *flags_out
 * (host), optionally clears it, and synchronises `stream`.  The registration loop carries the same word in its state
 * read-back (PIN_GN_STATE_STATUS), so Tracker.tracking hears about a flag without an extra read-back. */
#define PIN_STATUS_FP16_RANGE 1  /* pin_stage_decoder met a decoder parameter of magnitude >= 65504 or a non-finite one: the
                                  * split-fp16 image cannot hold it (mlp_h2.h); PIN_MLP=f32 selects the fp32 image */
int         pin_status(int32_t* flags_out, int32_t clear, void* stream);

/* host helper: cand_off_host[c] = (dx.primes) mod buffer_size, primes = (73856093,
 * 19349669, 83492791) (neural_points.py:82-84).  neighbor_dx_host is [n_cand][3]. */
int pin_candidate_offsets(const int32_t* neighbor_dx_host, int32_t n_cand, int64_t buffer_size,
                          int32_t* cand_off_host);

/* pos4[i] = (pos[i].xyz, bits(ts_create[i])) for i in [first, first+n) */
int pin_pack_positions(const float* pos, const int32_t* ts_create, int32_t first, int32_t n,
                       float* pos4, void* stream);

/* K1: full candidate search, reference API parity.  d2_out [n][Kc] f32, idx_out [n][Kc]
 * int64 global indices (-1 invalid) -- exactly the two tensors returned by
 * NeuralPoints.radius_neighborhood_search (neural_points.py:950-1009); global2local is
 * ignored here.  Used by query_certainty (neural_points.py:1011-1032). */
int pin_radius_search(const pin_search_params* sp, const float* query, int32_t n,
                      float* d2_out, int64_t* idx_out, void* stream);

/* K1+K2a: k nearest valid candidates per query, ascending (d2, candidate order).
 * `pose` (host, 12 floats row-major 3x4, may be NULL) is applied to the query points first
 * (transform_torch, utils/tools.py:534-553) and the transformed points are written to
 * query_out (may be NULL or alias nothing).  nbr_out [n][k][4] = (q - P).xyz and the bit
 * pattern of the neighbour index in the searched index space (-1 invalid, bit 30 set for
 * PIN_NONLOCAL neighbours); nn_count_out [n] counts valid candidates among ALL Kc
 * (neural_points.py:577). */
int pin_knn_query(const pin_search_params* sp, const float* query, int32_t n, int32_t k,
                  const float* pose_host, float* query_out, float* nbr_out,
                  int32_t* nn_count_out, void* stream);

/* ---- device-resident Gauss-Newton loop: no host round trip per iteration ---------------
 * pin_gn_state_init: state <- (T_init, last_res = 1e5, valid = 1, everything else 0).
 * pin_gn_knn: pin_knn_query / pin_knn_query_bricks (bc may be NULL) with the pose taken from
 *   the state; returns immediately on the device once the loop is done.
 * pin_gn_accumulate_solve: pin_gn_accumulate on the transformed points, then ONE wave sums the
 *   replicas, applies w /= 2 mean(w) (tracker.py:524), LM damping, the 6x6 float64 solve and
 *   expmap (tracker.py:656-679), T <- dT T and the reference's validity / convergence rules
 *   (tracker.py:147-184), and clears the sums for the next iteration (sums must be zeroed by
 *   the caller before the first one).  The host reads the state (PIN_GN_STATE_DOUBLES doubles) back once per frame.
 * pin_gn_loop_init: pin_gn_state_init with the pose passed by value (one launch, no copy) and the loop parameters stored IN the
 *   state (PIN_GN_STATE_LP).  On a state initialised this way pin_gn_accumulate_solve with the same parameters is TWO launches per
 *   iteration: the block of the tile kernel whose sums land last (a ticket in the state) runs the solve itself.  With other
 *   parameters, on a state from pin_gn_state_init, or for the decoder shapes without a tile kernel it launches the solve kernel
 *   behind the tile kernel as before; the results are the same bits either way. */
int pin_gn_state_init(double* state, const double* T_init_host, int32_t n_src, void* stream);
int pin_gn_loop_init(double* state, const double* T_init_host, int32_t n_src, const pin_gn_loop_params* lp, void* stream);
int pin_gn_knn(const pin_search_params* sp, const pin_brick_cache* bc, const float* src, int32_t n,
               int32_t k, const double* state, float* cur_out, float* nbr_out, int32_t* nn_count_out,
               void* stream);
/* pin_gn_knn through the brick cache, COHERENT across the iterations of one registration: the source points are searched
 * ~50 times under a pose that moves less and less.  iteration 0: the full search, which also leaves, per query, where it
 * stood, its k winners (entry offset | candidate number << 24; coh_win [n][8] uint32) and a margin -- how far it may move
 * before its voxel, the set of accepted candidates or the set of winners can change (coh_state [n][4] f32 = position,
 * margin squared).  iteration > 0: a query that stands within its margin re-measures only its winners and re-ranks them --
 * the same record, bit for bit, as the full search (its nn_count_out entry is left as it stands); the others take the full
 * search, which renews their state.  The state is valid for one (src, map, brick cache) and as long as nbr_out /
 * nn_count_out are the buffers of the previous iteration. */
int pin_gn_knn_coherent(const pin_search_params* sp, const pin_brick_cache* bc, const float* src, int32_t n, int32_t k,
                        const double* state, float* cur_out, float* nbr_out, int32_t* nn_count_out, float* coh_state,
                        uint32_t* coh_win, int32_t iteration, void* stream);
/* pin_gn_knn through the brick cache with per-query CANDIDATE LISTS kept across the iterations of one registration (the source
 * points are searched reg_iter_n times under a pose that moves by centimetres, tracker.py:114-184).  A full search leaves per
 * query its voxel and the entry offsets of its occupied candidate cells, compacted in candidate order: cell_state [n][4] int32 =
 * (voxel x, y, z, list length; -1 = this voxel needs the exact probe), cell_list [n][pin_knn_list_stride(n_cand)] uint32.  While
 * the query stays in that voxel a later call loads the list, gathers and measures those entries and runs the same tournament over
 * the shorter list; a query that has changed voxel rebuilds its list first.  Records and counts are the bits pin_gn_knn writes.
 * rebuild != 0 (the first iteration of a registration, or after ANY change of src / map / brick cache): every list is rebuilt. */
int32_t pin_knn_list_stride(int32_t n_cand);
int pin_gn_knn_listed(const pin_search_params* sp, const pin_brick_cache* bc, const float* src, int32_t n, int32_t k,
                      const double* state, float* cur_out, float* nbr_out, int32_t* nn_count_out, int32_t* cell_state,
                      uint32_t* cell_list, int32_t rebuild, void* stream);
/* the two halves of pin_gn_accumulate_solve, separately launchable (bench.py brackets the
 * accumulate kernel with HIP events) */
int pin_gn_accumulate_dev(const pin_field* f, const pin_gn_params* gp, const pin_color_term* color,
                          const float* cur, const float* nbr, const int32_t* nn_count, const float* sdf_labels,
                          int32_t n, double* sums, const double* state, void* stream);
int pin_gn_solve(double* sums, double* state, const pin_gn_loop_params* lp, void* stream);

int pin_gn_accumulate_solve(const pin_field* f, const pin_gn_params* gp, const pin_color_term* color,
                            const pin_gn_loop_params* lp, const float* cur, const float* nbr,
                            const int32_t* nn_count, const float* sdf_labels, int32_t n, double* sums,
                            double* state, void* stream);

/* Per-frame build of the brick cache (after reset_local_map / update).  counters_out:
 * int32[4] on the device = (bricks, entries, overflow flags, 0); flags != 0 or counts beyond
 * the capacities mean the caller must enlarge the buffers and rebuild. */
int64_t pin_brick_build_workspace_bytes(int32_t n_points, int32_t max_bricks);
int pin_brick_build(const pin_search_params* sp, const pin_brick_cache* bc, int32_t* counters_out,
                    void* stream);

/* pin_knn_query through the brick cache: same arguments, same outputs, same results. */
int pin_knn_query_bricks(const pin_search_params* sp, const pin_brick_cache* bc, const float* query,
                         int32_t n, int32_t k, const float* pose_host, float* query_out,
                         float* nbr_out, int32_t* nn_count_out, void* stream);

/* K2: NeuralPoints.query_feature tensor API (neural_points.py:590-746) from a kNN result.
 * feat_out: [n][11] (weighted_first) or [n][k][11]; weight_out [n][k]; certainty_out [n]
 * (NULL ok).  training != 0 applies the side effects of neural_points.py:685-710:
 * certainty_rw[idx] += w (float atomics), ts_update_rw[idx] = max(., query_ts). */
int pin_query_feature(const pin_field* f, const float* query, const float* nbr,
                      const int32_t* nn_count, int32_t n,
                      float* feat_out, float* weight_out, float* certainty_out,
                      int32_t training, float* certainty_rw, int32_t* ts_update_rw,
                      const int32_t* query_ts, void* stream);

/* K3: Decoder.sdf on caller-provided features (model/decoder.py:83-85): in [n][11] ->
 * out [n] (already multiplied by sdf_scale). */
int pin_decoder_sdf(const pin_field* f, const float* feat_in, int32_t n, float* sdf_out, void* stream);

/* Decoder.regress_color (model/decoder.py:112): sigmoid(mlp) of the 3 colour heads on
 * caller-provided features [n][11] -> color_out [n][3]. */
int pin_decoder_color(const pin_field* f, const float* feat_in, int32_t n, float* color_out, void* stream);

/* Colour branch of Tracker.query_source_points (utils/tracker.py:342-350) over the COLOUR
 * feature table fc->feats with fc->dec (3 heads): color_out [n][3] = regress_color (weighted
 * over the neighbours when weighted_first is off); value_out [n] = sum_c kappa[c]*color[c] and
 * grad_out [n][3] = its analytic gradient w.r.t. the query (kappa = (0.299, 0.587, 0.114) gives
 * the intensity used by registration, tools.py:408; a one-hot kappa gives one channel's
 * gradient).  Outputs may be NULL. */
int pin_color_query(const pin_field* fc, const float* query, const float* nbr, const int32_t* nn_count,
                    int32_t n, const float* kappa_host, float* color_out, float* value_out, float* grad_out,
                    void* stream);

/* K2+K3+K4 fused: interpolate, decode, analytic d sdf/d q through MLP, neighbour vectors
 * and IDW weights (Tracker.query_source_points, utils/tracker.py:297-354, get_gradient
 * utils/tools.py:247-260; Mesher.query_points, utils/mesher.py:60-140, with grad_out = NULL: forward sweep only).  Any
 * output pointer may be NULL.  With f->dec_image (pin_stage_decoder) the launch copies the staged decoder instead of
 * splitting it per block -- worth one staging launch when several queries follow on one decoder (the mesher's batches). */
int pin_sdf_query(const pin_field* f, const float* query, const float* nbr,
                  const int32_t* nn_count, int32_t n, float* sdf_out, float* grad_out,
                  float* std_out, float* certainty_out, void* stream);

/* K2..K5 fused: the same plus validity mask, Geman-McClure weights and the Gauss-Newton
 * normal-equation sums (tracker.py:409-524, 652-671).  sums_out: double[PIN_GN_NSUMS],
 * zeroed by this call.  sdf_labels may be NULL (all zero).  Per-point outputs optional.
 * color (NULL = off) adds the consistency weight or the photometric term of registration_step /
 * implicit_color_reg; sums_out is double[PIN_GN_REPLICAS][PIN_GN_NSUMS] (sum over replicas). */
int pin_gn_accumulate(const pin_field* f, const pin_gn_params* gp, const pin_color_term* color,
                      const float* query, const float* nbr, const int32_t* nn_count,
                      const float* sdf_labels, int32_t n, double* sums_out, float* sdf_out,
                      float* grad_out, void* stream);

/* Mapper.get_batch gathers (utils/mapper.py:482-488): rows `index[i]` of the sample pool
 * (coord_pool / sdf_label_pool / weight_pool / time_pool).  pool_weight / pool_ts / the
 * matching outputs may be NULL. */
int pin_gather_batch(const float* pool_coord, const float* pool_label, const float* pool_weight,
                     const int32_t* pool_ts, const int32_t* index, int32_t n, float* coord_out,
                     float* label_out, float* weight_out, int32_t* ts_out, void* stream);

/* All of Mapper.get_batch after its two torch.randint draws (utils/mapper.py:462-500) in one launch:
 * output row i < n_history is pool row index_history[i]; the remaining n - n_history rows are pool rows
 * new_idx[index_new_batch[i - n_history]] (the newly observed samples).  Indices are torch's int64 draws.
 * pool_color / color_out [..][color_channels] are optional (color_channels = 0).
 * query_out (optional): also write the training queries of pin_train_make_queries(coord_out, n, n_eik, decimation,
 * first, eps, query_out) in the same launch ([n + 6 n_eik][3]). */
int pin_gather_batch_drawn(const float* pool_coord, const float* pool_label, const float* pool_weight,
                           const int32_t* pool_ts, const float* pool_color, int32_t color_channels,
                           const int64_t* index_history, int32_t n_history, const int64_t* index_new_batch,
                           const int64_t* new_idx, int32_t n, float* coord_out, float* label_out,
                           float* weight_out, int32_t* ts_out, float* color_out, float* query_out,
                           int32_t n_eik, int32_t decimation, int32_t first, float eps, void* stream);

/* The same for n_batches batches in ONE launch (the iterations of a Mapper.mapping call: their index draws do not depend
 * on the training): batch b reads index_history + b * hist_stride and index_new_batch + b * new_stride and writes rows
 * [b * n, (b + 1) * n) of the outputs, queries [b * (n + 6 n_eik), ...).  One kNN launch over all the queries can follow:
 * the neural point positions do not change while the map trains. */
int pin_gather_batches_drawn(const float* pool_coord, const float* pool_label, const float* pool_weight,
                             const int32_t* pool_ts, const float* pool_color, int32_t color_channels,
                             const int64_t* index_history, int32_t n_history, const int64_t* index_new_batch,
                             const int64_t* new_idx, int32_t n, float* coord_out, float* label_out,
                             float* weight_out, int32_t* ts_out, float* color_out, float* query_out,
                             int32_t n_eik, int32_t decimation, int32_t first, float eps, int32_t n_batches,
                             int64_t hist_stride, int64_t new_stride, void* stream);

/* Neighbour records of the drawn samples out of records computed ONCE for every pool sample (pin_knn_query over the pool's
 * coordinates, pool_nbr [pool_n][k][4], pool_nn [pool_n]): the neural points do not move while the map trains
 * (utils/mapper.py:645-818 only updates features and decoder), so when the iterations of a Mapper.mapping call draw more
 * samples than the pool holds (batch 2^20: every pool sample ~6 times per call) the search of a sample is done once per
 * call and copied.  Batch b, sample i -> rows (b * q_per_batch + i) of nbr_out [.][k][4] / nn_out; the six Eikonal probes
 * of a sample are searched as before (they are not pool samples).  Index arrays as in pin_gather_batches_drawn. */
int pin_gather_records_drawn(const float* pool_nbr, const int32_t* pool_nn, int32_t k, const int64_t* index_history,
                             int32_t n_history, const int64_t* index_new_batch, const int64_t* new_idx, int32_t n,
                             int64_t q_per_batch, int32_t n_batches, int64_t hist_stride, int64_t new_stride,
                             float* nbr_out, int32_t* nn_out, void* stream);

/* K6a: query points of one training iteration: the batch itself followed by the six
 * central-difference points of every `decimation`-th sample (Mapper.get_numerical_gradient,
 * utils/mapper.py:682-686, 986-1008), grouped per sample: index n_main + 6*s + a with
 * a = x+, x-, y+, y-, z+, z- (the reference concatenates per axis; results are per-point
 * so the order is immaterial).  Sample s is coord[first + s*decimation]; `first` is 0 for a
 * whole batch and the phase of the shard when the batch is split over GPUs, so that the union
 * over shards is exactly the global coord[::decimation].  query_out: [n_main + 6*n_eik][3]. */
int pin_train_make_queries(const float* coord, int32_t n_main, int32_t n_eik, int32_t decimation,
                           int32_t first, float eps, float* query_out, void* stream);

/* expand = 1 for weighted_first, query_nn_k otherwise (one decode per neighbour) */
int64_t pin_train_workspace_bytes(int32_t n_queries, int32_t hidden, int32_t levels, int32_t expand);

/* K6: fused forward -> BCE-with-logits (utils/loss.py:45-63) + Eikonal (mapper.py:777-780)
 * -> backward over the queries produced by pin_train_make_queries and searched with
 * pin_knn_query.  Accumulates (+=, float atomics): feat_grad [M+1][8] (gradient of
 * local_geo_features), dec_grad [n_param] (flat, state_dict order; NULL = decoder frozen,
 * utils/tools.py:263-292).  Applies the training-mode side effects of query_feature for the
 * batch samples (certainty_rw += w, ts_update_rw = max(., sample_ts); neural_points.py:685-710).
 * loss_out: double[2] = (sum of BCE terms, sum of (|g|-1)^2) -- divide by the global counts.
 * pred_out (optional): sdf prediction of the batch samples [n_main].
 * Launch sequence (one stream): weighted_first, and per-neighbour decoding with a one-layer decoder: decoder image ->
 * fused tile kernel (gather, forward, loss, backward, feature scatter) -> streamed weight gradient -> finalize;
 * deeper per-neighbour decoders: forward / loss / backward / weight-gradient kernels over a unit-major workspace.
 * The workspace (pin_train_workspace_bytes) carries the operand stream of the weight gradient, slot copies of the
 * decoder gradient, per-block loss sums and the decoder image; its contents are scratch.
 * tp->eik_analytic (numerical_grad_on False, mapper.py:642-643, 677-678): the queries are the n_main batch samples alone
 * (pin_train_make_queries with n_eik = 0); the Eikonal term is taken on d pred / d q of every sample (tools.py:247-260)
 * and differentiated through it, as autograd's create_graph = True does; loss_out[1] = sum over the batch of
 * (|g|-1)^2; workspace of pin_train_workspace_bytes(2 * n_main, ...).  Built for weighted_first = 0 with a one-layer decoder; other shapes return an error. */
int pin_train_step(const pin_field* f, const pin_train_params* tp, const float* query,
                   const float* nbr, const int32_t* nn_count, const float* sdf_label,
                   const float* sample_weight, const int32_t* sample_ts, float* certainty_rw,
                   int32_t* ts_update_rw, float* feat_grad, float* dec_grad, double* loss_out,
                   float* pred_out, void* workspace, int64_t workspace_bytes, void* stream);

/* Where the last pin_train_step / pin_train_weight_grad of THIS thread that ran with tp->defer_dec_reduce left the decoder's gradient:
 * slots copies of n floats each (inside that call's workspace, valid until the next training launch on it) and the factor that
 * turns their sum into the gradient.  Returns -1 if that call did not defer (no decoder gradient asked for, or a path without
 * the slot copies). */
int pin_train_deferred_partial(const float** partial_out, int32_t* slots_out, int64_t* n_out, float* scale_out);


/* The second half of a pin_train_step called with tp->defer_weight_grad: streamed weight gradient into dec_grad (+=) and
 * the loss sums into loss_out, from the operand stream the first half left in `workspace` (same f / tp / workspace).  It
 * touches neither the feature table nor the feature gradients, so a caller may run it on ANOTHER stream beside the
 * optimiser launch that prepares the feature rows of the next iteration (pin_adam_lazy_prepare); the next
 * pin_train_step on this workspace, and the decoder's own step, must be ordered behind it. */
int pin_train_weight_grad(const pin_field* f, const pin_train_params* tp, float* dec_grad, double* loss_out,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* Colour term of one training iteration: regress_color on the colour features of the batch
 * (first n_main queries / kNN records of pin_train_step's query set), L1 loss on the surface
 * samples (color_diff_loss, utils/loss.py:31-42), backward into feat_grad (gradient of
 * local_color_features, +=) and dec_grad (colour decoder, +=; NULL = frozen).  loss_out:
 * double[1] = sum of (weighted) |pred - label| over surface samples and channels.  The workspace
 * needs pin_train_workspace_bytes(n_main, hidden, levels, expand) + 256 bytes.  Single GPU
 * (the surface-sample count of the mean is not all-reduced). */
int pin_train_color_step(const pin_field* fc, const pin_train_color_params* tp, const float* query,
                         const float* nbr, const int32_t* nn_count, const float* sdf_label,
                         const float* color_label, const float* sample_weight, float* feat_grad,
                         float* dec_grad, double* loss_out, void* workspace, int64_t workspace_bytes,
                         void* stream);

/* K7: torch.optim.Adam step (no amsgrad, no weight decay) as configured by setup_optimizer
 * (utils/tools.py:198-199): betas (0.9, 0.99), eps = adam_eps.  `step` counts from 1.
 * zero_grad != 0 clears grad in the same pass. */
int pin_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  int32_t step, float lr, float beta1, float beta2, float eps,
                  int32_t zero_grad, void* stream);

/* ---- K8..K10 map maintenance --------------------------------------------------------
 * All counts are produced on the device (int32) so calls can be chained without a host sync;
 * the host reads them back once per frame. */
int64_t pin_maint_workspace_bytes(int32_t n);

/* voxel_down_sample_torch (utils/tools.py:583-626): index of the point closest to its voxel
 * centre (distance quantised to 1000 levels, lowest index wins ties), one per voxel, ordered
 * by ascending linearised voxel id like torch.unique.  sel_out [<= n], count_out [1]. */
int pin_voxel_downsample(const float* points, int32_t n, float voxel_size, int32_t* sel_out,
                         int32_t* count_out, void* workspace, int64_t workspace_bytes, void* stream);
/* The same selection with a third of the launches: voxel id and tie-breaking value share one 64-bit sort key (the
 * winner of a voxel is then the first of its run).  The id has 64 - bits(1000 * 10^digits(n - 1)) bits for that: 37 for
 * a scan of up to 10^5 points, i.e. extents of up to 5 000 voxels per axis.  A point set that needs more is REPORTED --
 * count_out[0] = -1, sel_out undefined -- and the caller uses pin_voxel_downsample. */
int pin_voxel_downsample_fast(const float* points, int32_t n, float voxel_size, int32_t* sel_out,
                              int32_t* count_out, void* workspace, int64_t workspace_bytes, void* stream);

/* Spatial (Morton) order of a point set: out[i] = points[perm[i]], ascending 30-bit Morton code of floor(p / cell)
 * (10 bits per axis, wrapping).  Not a function of the reference: Tracker.tracking (utils/tracker.py:114-184) sums over
 * the source points, so their order is free, and the per-iteration kNN and the Gauss-Newton tile kernel run ~20 %
 * faster on a spatially coherent scan than on the down-sampler's x-fastest voxel order.  `out` must not alias
 * `points`; perm_out [n] optional.  Workspace: pin_maint_workspace_bytes(n). */
int pin_spatial_sort(const float* points, int32_t n, float cell, float* out, int32_t* perm_out, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* NeuralPoints.update (neural_points.py:334-416) for the down-sampled points points[sel[i]],
 * i < *n_sel: probe the hash table, decide which samples become new neural points, append
 * them (positions, packed mirror, identity orientation, timestamps, zero certainty) in sample
 * order and publish their indices in the table.  Feature rows of the new points are
 * initialised by the caller.  n_new_out [1] (device). */
int pin_map_update(const pin_map_arrays* ma, const pin_update_params* up, const float* points,
                   const int32_t* sel, const int32_t* n_sel, int32_t* n_new_out, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* NeuralPoints.reset_local_map (neural_points.py:445-511): travel-distance window + radius
 * masks, ordered compaction of the local arrays, global2local.  local_mask_out [n_points+1]
 * (bytes 0/1, last = 1), n_local_out [1] = M + 1 (the padding entry is counted). */
int pin_reset_local_map(const pin_map_arrays* ma, const pin_local_arrays* la, const pin_local_params* lp,
                        uint8_t* local_mask_out, int32_t* n_local_out, void* workspace,
                        int64_t workspace_bytes, void* stream);

/* NeuralPoints.assign_local_to_global (neural_points.py:515-526).  row_marker (or NULL): int32 [n_local], non-zero for the local
 * rows that changed since pin_reset_local_map cut the local map out of the global one -- only those are copied back (the others
 * still hold the global rows' bits).  Mapper.mapping passes the lazy optimiser's pending words: every row a training query of
 * the call read (features through the optimiser, certainty / ts_update through the queries' side effects). */
int pin_assign_local_to_global(const pin_map_arrays* ma, const pin_local_arrays* la, int32_t n_points,
                               int32_t n_local, const int32_t* row_marker, void* stream);

/* Exact sparse Adam for the feature tables.  The optimiser state is created anew by every
 * Mapper.mapping call (utils/mapper.py:615), so a row that no query of this call has touched yet
 * has g = m = v = 0 and its dense update is exactly zero; pin_adam_step_rows skips those rows and
 * is bit-identical to pin_adam_step over the whole table.  pin_mark_rows sets row_flags[idx] = 1
 * for every valid neighbour of the kNN records of an iteration ([n_records][4] floats); the caller
 * clears row_flags together with the optimiser state. */
int pin_mark_rows(const float* nbr, int64_t n_records, uint8_t* row_flags, void* stream);
int pin_adam_step_rows(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n_rows,
                       int32_t row_width, const uint8_t* row_flags, int32_t step, float lr, float beta1,
                       float beta2, float eps, int32_t zero_grad, void* stream);

/* Lazy exact Adam for the 8-wide feature tables (bit-identical to running pin_adam_step over the whole
 * table every iteration, as the reference does).  State: pending [rows] (int32, cleared when the optimiser is
 * reset; exp_avg / exp_avg_sq need no clearing), coef [2][t_max+1] =
 * lr/(1-beta1^t) and 1/sqrt(1-beta2^t) for t = 0..t_max (entry 0 unused), computed once on the host.
 * pin_adam_lazy_prepare, once per iteration t BEFORE the forward pass, over the kNN records of that iteration
 * ([n_records][4]): a row last read at iteration a takes the step it still owes (step a, with the gradient the
 * backward pass of iteration a left in `grad`, which is cleared), replays the gradient-free steps a+1 .. t-1 and
 * is marked as owing step t (one owner per row and call, elected by a compare-and-swap on its `pending` word; every
 * call of one optimiser lifetime must carry a different `step`).  pin_adam_lazy_flush (end of Mapper.mapping)
 * settles every touched row up to t_final.
 * A dense tensor (the decoder) can ride along: prepare(t) applies its step t-1, flush its step t_final
 * (= pin_adam_step with the table's coefficients and zero_grad).  With `image` (a decoder image of pin_stage_decoder for
 * this decoder: hidden x levels, out_dim heads) every updated parameter is written through to its image entries, so the
 * image stays current and pin_train_step (pin_train_params.dec_image_current) needs no staging launch per iteration. */
typedef struct pin_adam_dense {
    float* param; float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n;
    void* image;                       /* or NULL */
    int32_t hidden, levels, out_dim;   /* decoder shape of `image` */
    const float* grad_partial;         /* or NULL: the step's gradient is grad[e] + partial_scale * sum_c grad_partial[c * n + e], */
    int32_t partial_slots;             /*   c < partial_slots (what pin_train_step leaves with defer_dec_reduce: the same sum, in the */
    float partial_scale;               /*   same order, its own reduction launch would have added into grad) */
} pin_adam_dense;
int pin_adam_lazy_prepare(const float* nbr, int64_t n_records, float* param, float* grad, float* exp_avg,
                          float* exp_avg_sq, int32_t* pending, int32_t step, const float* coef, int32_t t_max,
                          float beta1, float beta2, float eps, const pin_adam_dense* dense, void* stream);
/* pin_adam_lazy_prepare for LARGE batches (many more records than rows; a 2^20 batch: 13 M records over 2.2 M rows): the
 * records only flag their rows (row_flags [n_rows] uint8, all zero on entry and again on return), one pass over the rows
 * settles the flagged ones -- no election, the same arithmetic, the same bits. */
int pin_adam_lazy_prepare_rows(const float* nbr, int64_t n_records, float* param, float* grad, float* exp_avg,
                               float* exp_avg_sq, int32_t* pending, uint8_t* row_flags, int64_t n_rows, int32_t step,
                               const float* coef, int32_t t_max, float beta1, float beta2, float eps,
                               const pin_adam_dense* dense, void* stream);
int pin_adam_lazy_flush(float* param, float* grad, float* exp_avg, float* exp_avg_sq, const int32_t* pending,
                        int64_t n_rows, int32_t t_final, const float* coef, int32_t t_max, float beta1, float beta2,
                        float eps, const pin_adam_dense* dense, void* stream);

/* A GROUP of consecutive training iterations in one call (Mapper.mapping gathers and searches a group's batches in one launch each:
 * their inputs sit in strided buffers): per iteration i = 0 .. n_iters - 1 exactly the launches of
 *     pin_adam_lazy_prepare[_rows](records of iteration i, step first_step + i, dense rider = the decoder with the previous
 *                                  iteration's weight gradient as slot copies)
 *     pin_train_step(tp with defer_dec_reduce = 1, except on the call's last iteration)
 * -- the host-side loop of engine.MapTrainer.step_batch moved behind the ABI: one foreign call per group instead of two or three
 * per iteration (0.57 -> 0.35 ms of host time per Mapper.mapping call at 12 iterations).  Strides are in ELEMENTS of the
 * pointer's type, from one iteration to the next.  partial_*: in = the slot copies the iteration BEFORE the group left (NULL: none),
 * out = what the group's last iteration left (NULL after the call's last iteration). */
typedef struct pin_train_group {
    int32_t n_iters, first_step;
    int32_t last_of_call;                 /* the group's last iteration is the Mapper.mapping call's last */
    int32_t rows_form;                    /* pin_adam_lazy_prepare_rows instead of pin_adam_lazy_prepare */
    const float* query;   int64_t query_stride;     /* [n_iters][Q][3] */
    const float* nbr;     int64_t nbr_stride;       /* [n_iters][Q][k][4] */
    const int32_t* nn;    int64_t nn_stride;        /* [n_iters][Q] */
    const float* sdf_label; int64_t label_stride;   /* [n_iters][n_main] */
    const float* sample_weight; int64_t weight_stride;  /* or NULL */
    const int32_t* sample_ts;   int64_t ts_stride;      /* or NULL */
    float* certainty_rw; int32_t* ts_update_rw;     /* or NULL (side effects deferred) */
    float* feat_grad; float* dec_grad; double* loss_out;
    void* workspace; int64_t workspace_bytes;
    int64_t n_records;                    /* Q * k: records per iteration */
    float* exp_avg; float* exp_avg_sq; int32_t* pending; uint8_t* row_flags; int64_t n_rows;   /* lazy Adam of f->feats */
    const float* coef; int32_t t_max; float beta1, beta2, eps;
    pin_adam_dense dense;                 /* the decoder as the rider (grad_partial is set per iteration from partial_*) */
    const float* partial; int32_t partial_slots; float partial_scale;   /* in / out, see above */
    /* ABI v18 -- the TWO-STREAM form (engine.MapTrainer.step_batch's `overlap` path, the default with a colour decoder) and the
     * colour branch of the iteration (mapper.py:668-671, 802-812).  side_stream != NULL: per iteration the lazy launch carries no
     * rider, pin_train_step stops behind the tile kernel (defer_weight_grad), pin_train_weight_grad and the decoder's step (`dense`
     * alone: pin_adam_lazy_flush with no table) run on side_stream behind it, and the NEXT tile kernel waits for them; partial_*
     * unused.  The caller orders the two streams in front of the group (side work of an earlier call) and behind it (the last
     * iteration's side work is still running when the call returns).  fc != NULL (either form): on `stream`, behind the above, the
     * colour table's lazy launch (c_dense riding along unless its param is NULL: a frozen colour decoder) and pin_train_color_step
     * on the same queries / records; with a colour branch the in-line form reduces every iteration's weight gradient itself.
     * dense.param == NULL (in-line form only): a FROZEN decoder (freeze_decoders, utils/tools.py:263-292 -- every frame of a run
     * after freeze_after_frame): no rider, no weight gradient, dec_grad unused. */
    void* side_stream;
    const pin_field* fc; const pin_train_color_params* cp;
    const float* color_label; int64_t color_stride;   /* [n_iters][n_main][3] */
    float* c_feat_grad; float* c_dec_grad; double* c_loss_out;
    void* c_workspace; int64_t c_workspace_bytes;
    float* c_exp_avg; float* c_exp_avg_sq; int32_t* c_pending; uint8_t* c_row_flags;   /* lazy Adam of fc->feats (n_rows rows) */
    pin_adam_dense c_dense;
} pin_train_group;
int pin_train_group_steps(const pin_field* f, const pin_train_params* tp, pin_train_group* g, void* stream);

/* ---- Mapper.process_frame data path (utils/mapper.py:162-449) ------------------------------ */

/* Bytes of workspace for the pool kernels on n elements (pin_new_sample_index needs n more). */
int64_t pin_pool_workspace_bytes(int64_t n);

/* K12: DataSampler.sample (utils/data_sampler.py:18-260) fused with the pool append and the
 * sensor->world transform of Mapper.process_frame (utils/mapper.py:275-300).  points: n scan
 * points in the sensor frame, rows of `row_stride` floats (xyz first); colors: pointer to the
 * first colour channel of row 0 (same stride) or NULL.  rnd_surface [surface_n*n] ~ N(0,1),
 * rnd_front [front_n*n], rnd_behind [behind_n*n] ~ U[0,1): the reference's torch.randn / rand
 * draws in its order of generation (entry s*n + i belongs to ray i).  Writes n*A samples
 * (A = 1 + surface_n + front_n + behind_n) ray-wise: sample j of ray i at i*A + j. */
int pin_sample_rays(const pin_sample_params* p, const float* points, const float* colors, int32_t row_stride,
                    int32_t n, const float* rnd_surface, const float* rnd_front, const float* rnd_behind,
                    const pin_pool_arrays* out, void* stream);

/* K13a: distance window of the pool filter (utils/mapper.py:303-310): mask[i] =
 * ||global_coord[i] - origin||^2 < radius^2 evaluated in float64; count_out [1] = kept;
 * true_index [n] (optional) = torch.nonzero(mask) for the random discard. */
int pin_pool_window_mask(const float* global_coord, int32_t n, const double* origin, double radius,
                         uint8_t* mask, int32_t* true_index, int32_t* count_out, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* K13b: filter_mask[true_indices[discarded_index]] = False (utils/mapper.py:314-323);
 * discard_index are the caller's torch.randint draws. */
int pin_pool_discard(uint8_t* mask, const int32_t* true_index, const int64_t* discard_index,
                     int32_t n_discard, void* stream);

/* K13c: pool[filter_mask] for all pool arrays at once, order preserved, out of place
 * (src -> dst).  counts_out [2] = {kept, kept among the last n_cur elements}
 * (pool_sample_count, cur_sample_count of utils/mapper.py:338-346). */
int pin_pool_compact(const pin_pool_arrays* src, const pin_pool_arrays* dst, const uint8_t* mask, int32_t n,
                     int32_t n_cur, int32_t* counts_out, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* K14: NeuralPoints.query_certainty (model/neural_points.py:1011-1033): max over the valid
 * candidates (sp->cand_off, sp->max_valid_dist2; no time filter) of certainty[global index],
 * 0 if none.  process_frame calls it with the own-cell neighbourhood (1 candidate). */
int pin_query_certainty(const pin_search_params* sp, const float* certainty, const float* query, int32_t n,
                        float* certainty_out, void* stream);

/* K14b: new_idx = where(certainty < certainty_thre & |sdf_label| < label_thre) + offset
 * (utils/mapper.py:405-416), ascending.  index_out [<= n] int64, count_out [1]. */
int pin_new_sample_index(const float* certainty, const float* sdf_label, int32_t n, float certainty_thre,
                         float label_thre, int64_t offset, int64_t* index_out, int32_t* count_out,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* update_points of process_frame (utils/mapper.py:236-246): rows[i] ([n][3]) of the samples with
 * |sdf_label[i]| < label_thre (= surface_sample_range_m * map_surface_ratio), order preserved.
 * out [<= n][3], count_out [1]. */
int pin_select_surface_points(const float* rows, const float* sdf_label, int32_t n, float label_thre,
                              float* out, int32_t* count_out, void* workspace, int64_t workspace_bytes,
                              void* stream);

/* transform_torch (utils/tools.py:534-553): out[i] = R * points[i] + t in float32 with the pose
 * rows [R|t] given as 12 doubles (cast to float32 like `transformation.to(points)`).  points
 * rows have `row_stride` floats (xyz first); out [n][3]. */
int pin_transform_points(const float* points, int32_t row_stride, int32_t n, const double* pose, float* out,
                         void* stream);

/* transform_batch_torch (utils/tools.py:556-580), in place: points[i] <- R_f p + t_f with f = frame[i] and
 * T [n_frames][12] float32 rows [R|t] (NeuralPoints.adjust_map neural_points.py:791-817 with f = ts_create,
 * Mapper.transform_data_pool utils/mapper.py:527-531 with f = time_pool).  quat / dquat (optional): the
 * orientation update of adjust_map, quat[i] <- quat_multiply(dquat[f], quat[i]) (w, x, y, z). */
int pin_transform_by_frame(float* points, int32_t n, const int32_t* frame, const float* T, int32_t n_frames,
                           float* quat, const float* dquat, void* stream);

/* Row gather out[i] = src[index[i]] for pools pin_gather_batch does not cover (color_pool,
 * utils/mapper.py:494-495). */
int pin_gather_rows(const float* src, int32_t width, const int32_t* index, int32_t n, float* out, void* stream);

/* ---- SLAMDataset.preprocess_frame data path (dataset/slam_dataset.py:359-505) ----------------
 * The two voxel_down_sample_torch passes are pin_voxel_downsample + pin_gather_rows. */

/* crop_frame (dataset/slam_dataset.py:1229-1247): rows ([n][width], xyz first) with
 * min_range < ||xyz|| < max_range and min_z < z < max_z, order preserved, with their
 * per-point timestamps (ts / ts_out may be NULL).  count_out [1]; workspace
 * pin_pool_workspace_bytes(n) + n bytes. */
int pin_crop_frame(const float* points, int32_t width, int32_t n, const float* ts, float min_z, float max_z,
                   float min_range, float max_range, float* points_out, float* ts_out, int32_t* count_out,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* intrinsic_correct (dataset/slam_dataset.py:1251-1269), in place on rows of `width` floats. */
int pin_intrinsic_correct(float* points, int32_t width, int32_t n, double correct_deg, void* stream);

/* deskewing (utils/tools.py:747-779), in place: p_i <- exp(t_i log R) p_i + t_i trans with
 * t_i = (ts_i - min ts)/(max ts - min ts) - ts_mid_pose and pose = T_last<-cur as 16 doubles
 * (row-major 4x4).  workspace >= 64 bytes. */
int pin_deskew(float* points, int32_t width, int32_t n, const float* ts, const double* pose, double ts_mid_pose,
               void* workspace, int64_t workspace_bytes, void* stream);

/* ---- data-parallel mapper collectives ---------------------------------------------
 * The reference trains on one GPU (pin_slam.py:8).  Mapper.mapping optimises the local feature
 * table AND the decoder (neural_points.parameters() + decoder parameters, utils/mapper.py:604;
 * setup_optimizer utils/tools.py:153-203), so a batch sharded over ranks needs the SUM of the
 * per-rank gradients of both before the (replicated) Adam step, and the training-mode side effects of
 * query_feature (certainty += w, ts_update = max; neural_points.py:660-683) merged once per mapping call.
 * RCCL (xGMI) is bound at run time: pin_comm_load dlopens the librccl the host process already uses. */
#define PIN_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */

/* dlopen RCCL (rccl_path, else librccl.so.1 / librccl.so on the loader path).  Idempotent. */
int pin_comm_load(const char* rccl_path);
/* HOST: rank 0 fills id_out_host[PIN_COMM_ID_BYTES] (ncclGetUniqueId); the caller hands it to the other ranks. */
int pin_comm_unique_id(void* id_out_host);
/* HOST, collective over `world` processes (one per GPU, current device = the rank's GPU): ncclCommInitRank.
 * *comm_out_host is an opaque communicator handle. */
int pin_comm_init_rank(const void* id_host, int32_t rank, int32_t world, void** comm_out_host);
int pin_comm_destroy(void* comm);

/* In-place SUM all-reduce of the flat fp32 gradient buffer [decoder grads | feature grads] on `stream`
 * (ncclAllReduce, ncclFloat32, ncclSum): one call per Mapper.mapping iteration, between the backward pass
 * (pin_train_step) and the optimiser step (pin_adam_step). */
int pin_allreduce_grads(void* comm, float* grads, int64_t count, void* stream);

/* Out-of-place SUM all-reduce of `count` floats on `stream` (ncclAllReduce; send == recv allowed).  Used by the
 * spatially sharded mapper for the compact [decoder | halo rows] gradient buffer and for the owner merge. */
int pin_allreduce_f32(void* comm, const float* send, float* recv, int64_t count, void* stream);

/* The data half of SLAMDataset.preprocess_frame (dataset/slam_dataset.py:359-505) in ONE call with the stage counts on the
 * device: voxel_down_sample_torch at train resolution (utils/tools.py:583-626) -> crop_frame (slam_dataset.py:1229-1247) ->
 * [intrinsic_correct, :1251-1269] -> voxel_down_sample_torch at source resolution -> deskewing of the source (utils/tools.py:
 * 747-779).  Every stage reads its input count from device memory (launches are sized by the raw scan), so nothing comes back
 * to the host in between.  scan [n][width] rows (xyz first), ts [n] or NULL.  Outputs, sized for n rows by the caller:
 * pc_out [.][width] the cropped cloud (rows 0 .. c2), ts_out its timestamps, source_xyz_out [.][3] the registration source
 * (rows 0 .. c3; deskewed with `pose` when deskew != 0 and ts is given), source_rest_out [.][width - 3] its other columns
 * (may be NULL), counts_out (DEVICE int32[3]) = c1 points kept by the first down-sampling, c2 by the crop, c3 by the second
 * down-sampling.  c1 or c3 = -1: that point set's voxel ids do not fit the one-word sort key (pin_voxel_downsample_fast) --
 * the caller runs the stages one by one instead.  Results are those of the separate entry points, bit for bit. */
typedef struct pin_preprocess_params {
    float train_vox, source_vox;             /* vox_down_m, source_vox_down_m (scaled by the adaptive range when that is on) */
    float min_z, max_z, min_range, max_range;/* crop_frame */
    double correct_deg;                      /* kitti_correction_on ? correction_deg : 0 */
    int32_t want_source, deskew;
    double pose[16];                         /* last_odom_tran, row-major (deskewing) */
    double ts_mid_pose;
} pin_preprocess_params;
int64_t pin_preprocess_workspace_bytes(int32_t n, int32_t width);
int pin_preprocess_frame(const pin_preprocess_params* pp, const float* scan, int32_t width, int32_t n, const float* ts,
                         float* pc_out, float* ts_out, float* source_xyz_out, float* source_rest_out, int32_t* counts_out,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* ---- spatially sharded data-parallel mapper -------------------------
 * xGMI is point-to-point, so the exchange is made small instead of fast: the voxel grid is cut into `world`
 * axis-aligned boxes (a k-d split of the drawn batch, computed by the host and identical on every rank) and rank r
 * trains on the samples of every drawn batch whose voxel lies in box r.  A neural point whose voxel is more than
 * `reach` cells inside its box can only be a neighbour of that box's samples: its gradient is complete on the
 * owner, its Adam step is the owner's (lazy, as on one GPU) and nobody else reads it during the call.  The other
 * rows -- the HALO, a few per cent of the map -- are kept identical on every rank: per iteration ONE all-reduce of
 * the compact buffer [decoder grads | halo-row grads] and the same dense Adam step on all of them.  At the end of
 * Mapper.mapping every rank publishes the rows it owns (pin_dp_owner_pack + one all-reduce, once per call).
 * The sum over ranks of the per-rank gradients equals the single-GPU gradient of the drawn batch whatever the
 * partition is (losses are normalised by the GLOBAL counts); the Eikonal sub-sample stays the global coord[::dec]. */
typedef struct pin_dp_regions {
    const int32_t* boxes;   /* DEVICE [world][6]: lo x,y,z / hi x,y,z in voxel coordinates floor(p / resolution), hi
                             * exclusive; unbounded faces are INT32_MIN / INT32_MAX.  The boxes tile the grid. */
    int32_t world, rank;
    int32_t reach;          /* cells a training query of a sample can lie from the sample's voxel plus the search radius:
                             * num_nei_cells + ceil(eik_eps / resolution) (0 without Eikonal probes) */
    float resolution;       /* voxel_size_m (the grid of the neighbour search, neural_points.py:963) */
} pin_dp_regions;

/* HOST (no device work): the boxes.  boxes_out_host [world][6] from n sample voxels cells_host [n][3]: recursive
 * splits along the axis of largest extent at the voxel coordinate that divides the samples in proportion to the
 * ranks on either side (std::nth_element).  Deterministic in its input: every rank calls it on the same sub-sample
 * of the same drawn batch and gets the same boxes without an exchange. */
int pin_dp_kd_boxes(const int32_t* cells_host, int32_t n, int32_t world, int32_t* boxes_out_host);

/* Rank 0's boxes become everybody's (a rank whose host computed something else -- a pool that differs in one bit -- must
 * not run with a different halo: the all-reduce sizes would disagree): every int32 coordinate travels as two fp32 halves
 * (v >> 16 and v & 0xffff, both exact in fp32), rank 0 sends them and the others send zeros through pin_allreduce_f32;
 * this decodes halves [world][6][2] into boxes_out [world][6] on the device. */
int pin_dp_boxes_decode(const float* halves, int32_t world, int32_t* boxes_out, void* stream);

/* Replica-consistency check of a spatially sharded call (no counterpart in the reference, which trains on one GPU:
 * pin_slam.py:8).  Only the boxes are agreed through an exchange; halo list, owner lists and the partition are derived by every
 * rank from ITS copy of the map and the pool, and pin_dp_rows_unpack / the all-gather counts index those local lists with remote
 * payloads -- replicas that differ in one bit would corrupt the map silently.  sig_out [2 * (world + 4 + n_counts)] fp32:
 * every 32-bit word as its two 16-bit halves (exact in fp32, and their SUM over <= 64 ranks is exact): word 0 = *n_halo_dev,
 * words 1 .. world + 1 = offsets[0 .. world] of pin_dp_owner_lists, word world + 2 = checksum of halo_rows[0 .. *n_halo_dev),
 * word world + 3 = checksum of the low words of first_batch[0 .. n_first) (the drawn indices of the first iteration), then
 * counts[0 .. n_counts) (this rank's per-iteration sample counts of pin_dp_partition).  The caller all-reduces the buffer:
 * the first world + 4 words must come back as world x its own on every rank, the counts must add up to the batch. */
int pin_dp_signature(const int32_t* halo_rows, const int32_t* n_halo_dev, const int32_t* offsets, int32_t world,
                     const int64_t* first_batch, int32_t n_first, const int32_t* counts, int32_t n_counts, float* sig_out,
                     void* stream);

/* Voxel coordinates of every `stride`-th sample of one drawn batch (positions s*stride < n of the batch that
 * pin_gather_batch_drawn would gather): cells_out [n_out][3] int32 -- what the host cuts its k-d boxes from. */
int pin_dp_sample_cells(const float* pool_coord, const int64_t* index_history, int32_t n_history,
                        const int64_t* index_new_batch, const int64_t* new_idx, int32_t n, int32_t stride,
                        int32_t n_out, float resolution, int32_t* cells_out, void* stream);

/* Which samples of the drawn batches are this rank's: for batch b < n_batches, sel_out[b][0 .. counts[b][0]) = the
 * positions i < n of the batch whose sample lies in box `rank` (any order), eik_sel_out[b][0 .. counts[b][1]) those
 * of them with i % decimation == 0 (the Eikonal sub-sample coord[::dec] of the GLOBAL batch, mapper.py:683).
 * counts_out [n_batches][2] is cleared here; an entry larger than cap / eik_cap means the lists are truncated
 * (grow and call again).  Index arrays as in pin_gather_batches_drawn.  pool_region [pool_rows] (scratch, one byte per
 * pool row): the box of every row of pool_coord is computed once, the draws then look it up.  pool_label /
 * surface_counts_out (may be NULL): surface_counts_out[b] = samples of the WHOLE batch b with |label| < surface_range --
 * the normalisation of the colour loss, which a rank cannot count from its own samples. */
int pin_dp_partition(const pin_dp_regions* rg, const float* pool_coord, const int64_t* index_history, int32_t n_history,
                     const int64_t* index_new_batch, const int64_t* new_idx, int32_t n, int32_t decimation,
                     int32_t n_batches, int64_t hist_stride, int64_t new_stride, int32_t* sel_out, int32_t cap,
                     int32_t* eik_sel_out, int32_t eik_cap, int32_t* counts_out, int64_t pool_rows, uint8_t* pool_region,
                     const float* pool_label, float surface_range, int32_t* surface_counts_out, void* stream);

/* Mapper.get_batch for the selected samples of n_batches drawn batches (b-th batch: sel / eik_sel / counts rows
 * b0 + b of pin_dp_partition's outputs): *_out[b][j] = pool row of batch position sel[b][j], j < counts[b][0];
 * query_out [n_batches][cap + 6 eik_cap][3]: the samples, then (from row counts[b][0]) the six +-eps probes of every
 * Eikonal sample in the order x+, x-, y+, y-, z+, z-. */
int pin_dp_gather(const float* pool_coord, const float* pool_label, const float* pool_weight, const int32_t* pool_ts,
                  const float* pool_color, int32_t color_channels, const int64_t* index_history, int32_t n_history,
                  const int64_t* index_new_batch, const int64_t* new_idx, int64_t hist_stride, int64_t new_stride,
                  const int32_t* sel, int32_t cap, const int32_t* eik_sel, int32_t eik_cap, const int32_t* counts,
                  int32_t n_batches, float* coord_out, float* label_out, float* weight_out, int32_t* ts_out,
                  float* color_out, float* query_out, float eps, void* stream);

/* Training-mode side effects of a whole Mapper.mapping call at once (query_feature(training_mode=True), neural_points.py:
 * 685-710: certainty[idx_k] += w_k and ts_update[idx_k] = max(., sample ts) for every main query).  With per-pool-sample
 * neighbour records (pin_gather_records_drawn / pin_dp_gather_records) a sample drawn m times adds m x the same weights:
 * pin_count_draws counts how often every pool row appears in the call's index batches (index_history [n_history_total] and
 * new_idx[index_new_batch[.]] [n_new_total], all iterations; count [pool rows] zeroed by the caller), and
 * pin_certainty_from_records applies certainty[idx_k] += m w_k / the ts maximum from the records in one pass over the pool rows
 * (pool_to_rec: row -> record index, or NULL for the identity; rows with pool_to_rec < 0 are another rank's).  The training
 * launches of such a call run with certainty_rw = ts_update_rw = NULL. */
int pin_count_draws(const int64_t* index_history, int64_t n_history_total, const int64_t* index_new_batch, const int64_t* new_idx,
                    int64_t n_new_total, int32_t* count, void* stream);
int pin_certainty_from_records(const float* rec_nbr, const int32_t* rec_nn, int32_t k, const int32_t* pool_to_rec,
                               const int32_t* count, const int32_t* pool_ts, int64_t n_pool, float* certainty_rw,
                               int32_t* ts_update_rw, void* stream);

/* Neighbour records of THIS rank's pool samples, searched once per Mapper.mapping call (the neural points do not move while
 * the map trains; a 2^20 batch draws every pool sample ~6 times per call): the one-GPU path's pin_gather_records_drawn for
 * the spatial shards.
 * pin_dp_own_pool: after pin_dp_partition (which wrote pool_region): the pool rows of box `rank`, in pool order --
 *   own_coord_out [own_cap][3] their coordinates (the queries of ONE search over them), pool_to_own_out [pool_rows] = position
 *   in that list (-1 for the other boxes' rows), *count_out (device) their number; workspace: 4 bytes per 256 pool rows + 256.
 * pin_dp_gather_records: for batch b < n_batches and this rank's sample j < counts[b][0]: record and neighbour count of the
 *   sample's pool row -> slot b of the trainer's record buffers (stride cap + 6 * ecap queries per batch, as pin_dp_gather lays
 *   the queries out); the Eikonal probes behind them are searched per iteration as before. */
int pin_dp_own_pool(const uint8_t* pool_region, int32_t pool_rows, int32_t rank, const float* pool_coord, float* own_coord_out,
                    int32_t own_cap, int32_t* pool_to_own_out, int32_t* count_out, void* workspace, int64_t workspace_bytes,
                    void* stream);
int pin_dp_gather_records(const float* rec_nbr, const int32_t* rec_nn, int32_t k, const int32_t* pool_to_own,
                          const int64_t* index_history, int32_t n_history, const int64_t* index_new_batch, const int64_t* new_idx,
                          int64_t hist_stride, int64_t new_stride, const int32_t* sel, int32_t cap, int32_t ecap,
                          const int32_t* counts, int32_t n_batches, float* nbr_out, int32_t* nn_out, void* stream);

/* Halo of the partition over the feature rows at pos [n_rows][3]: halo_rows_out = ascending indices of the rows
 * that are NOT at least `reach` cells inside their own box (the same list on every rank), *count_out of them (at
 * most halo_cap are written); owner_out [n_rows] = box of every row's voxel, | 0x80 for the halo rows; lazy_pending (may be NULL) [n_rows]:
 * halo rows <- PIN_ADAM_ROW_EXCLUDED so that pin_adam_lazy_* leave them to pin_dp_halo_adam.
 * Workspace: pin_maint_workspace_bytes(n_rows). */
int pin_dp_mark_halo(const pin_dp_regions* rg, const float* pos, int32_t n_rows, int32_t* halo_rows_out,
                     int32_t halo_cap, int32_t* count_out, uint8_t* owner_out, int32_t* lazy_pending, void* workspace,
                     int64_t workspace_bytes, void* stream);

/* After the backward pass: packed_out[h][0..8) <- feat_grad[halo_rows[h]][0..8), and the rows are cleared. */
int pin_dp_halo_pack(const int32_t* halo_rows, int32_t n_halo, float* feat_grad, float* packed_out, void* stream);

/* After the all-reduce: the dense Adam step `step` on the halo rows with the summed gradients grad_sum [n_halo][8];
 * exp_avg / exp_avg_sq [n_halo][8] are the compact moments (zero at the start of a Mapper.mapping call); coef =
 * the step coefficient table of pin_adam_lazy_* ([2][t_max + 1]).  Identical inputs on every rank -> identical rows. */
int pin_dp_halo_adam(const int32_t* halo_rows, int32_t n_halo, float* feats, const float* grad_sum, float* exp_avg,
                     float* exp_avg_sq, int32_t step, const float* coef, int32_t t_max, float beta1, float beta2,
                     float eps, void* stream);

/* End of the call, form 1 (any transport with an all-reduce): out[row] = (owner[row] & 0x7f) == rank ? feats[row] : 0 over
 * n_rows rows of 8 floats; the SUM over ranks (pin_allreduce_f32 into the feature table) is the trained table on every
 * rank.  owner[] as pin_dp_mark_halo writes it: box | 0x80 for halo rows. */
int pin_dp_owner_pack(const uint8_t* owner, int32_t rank, const float* feats, int32_t n_rows, float* out, void* stream);

/* End of the call, form 2 (half the bytes: an all-GATHER of what each rank owns instead of an all-reduce of the table).
 * pin_dp_owner_lists: lists_out = the PRIVATE rows (no halo bit) of box 0, then of box 1, ... (a deterministic counting
 * sort of owner[]: the same lists on every rank), offsets_out [world + 1] (device) where each list starts.
 * pin_dp_rows_pack: out[j] = (8 features, certainty, ts_update bits[, 8 colour features when color_feats != NULL]) of row
 * rows[j] -- a rank packs ITS list, padded by the caller to the longest list (segment_rows); pin_allgather_f32
 * (ncclAllGather, count_per_rank = 10 (18 with colour) * segment_rows floats);
 * pin_dp_rows_unpack: the other ranks' records back into the tables.  A private row is only ever touched by its owner's
 * samples -- features, certainty and ts_update alike -- so the owner's values ARE the merged values; the halo rows'
 * certainty (sum of deltas) and ts_update (max) go through pin_dp_sync_side_effects in compact form
 * (pin_dp_halo_side_gather / _scatter), their features are identical everywhere already. */
int64_t pin_dp_owner_lists_workspace_bytes(int32_t n_rows, int32_t world);
int pin_dp_owner_lists(const uint8_t* owner, int32_t n_rows, int32_t world, int32_t* lists_out, int32_t* offsets_out,
                       void* workspace, int64_t workspace_bytes, void* stream);
int pin_dp_rows_pack(const int32_t* rows, int32_t count, const float* feats, const float* certainty, const int32_t* ts_update,
                     const float* color_feats, float* out, void* stream);
int pin_dp_rows_unpack(const float* gathered, int32_t segment_rows, const int32_t* lists, const int32_t* offsets, int32_t world,
                       int32_t rank, float* feats, float* certainty, int32_t* ts_update, float* color_feats, void* stream);
/* lazy_pending[rows[h]] <- PIN_ADAM_ROW_EXCLUDED: the halo rows of a second lazily stepped table (the colour features). */
int pin_dp_exclude_rows(const int32_t* rows, int32_t n, int32_t* lazy_pending, void* stream);
int pin_dp_halo_side_gather(const int32_t* halo_rows, int32_t n_halo, const float* certainty, const float* certainty0,
                            const int32_t* ts_update, float* cert_out, float* cert0_out, int32_t* ts_out, void* stream);
int pin_dp_halo_side_scatter(const int32_t* halo_rows, int32_t n_halo, const float* cert_in, const int32_t* ts_in,
                             float* certainty, int32_t* ts_update, void* stream);
/* ncclAllGather of count_per_rank floats per rank on `stream` (recv [world][count_per_rank]). */
int pin_allgather_f32(void* comm, const float* send, float* recv, int64_t count_per_rank, void* stream);

/* ---- post-loop map maintenance on the device ------------------------------------ */

/* NeuralPoints.recreate_hash (model/neural_points.py:819-908) with voxel_down_sample_min_value_torch
 * (utils/tools.py:629-668) inside: per voxel of size `resolution` the index of the point with the smallest value
 * (|ts - cur_ts| or max(certainty) - certainty, quantised to 1000 levels, lowest index on ties; the reference's
 * `grid.max()` stride of the voxel id is reproduced), sel_out [<= n_points] in voxel-id order, *count_out of them.
 * dst == NULL (kept_points = True): the table is cleared and re-filled with table[hash(pos[sel[r]])] = sel[r];
 * when several samples share a slot the last in sample order stays (the reference's index_put_ leaves that
 * unspecified).  dst != NULL (kept_points = False, the final merge): rows sel[r] of every array of `src` are
 * gathered to row r of `dst` (feature padding row appended, pos4 mirror written) and the table, which both structs
 * share, is rebuilt over the new indices.  Workspace: pin_maint_workspace_bytes(n_points) + 4 * n_points bytes. */
int pin_hash_rebuild(const pin_map_arrays* src, const pin_map_arrays* dst, const pin_rehash_params* rp,
                     int32_t* sel_out, int32_t* count_out, void* workspace, int64_t workspace_bytes, void* stream);

/* NeuralPoints.prune_map (model/neural_points.py:748-789): rows with certainty < certainty_thre (and, unless
 * global_prune, |travel[cur_ts] - travel[ts_update]| > diff_travel_dist_local) are dropped; the kept rows of `src`
 * are written to `dst` in order (ordered compaction, feature padding row appended, pos4 mirror written);
 * *n_keep_out = rows kept.  dst == NULL: count only (the caller decides from n_points - n_keep > min_prune_count, as the
 * reference does, before it spends a second set of map arrays on the compaction).
 * Workspace: pin_maint_workspace_bytes(n_points). */
int pin_prune_map(const pin_map_arrays* src, const pin_map_arrays* dst, const pin_prune_params* pp,
                  int32_t* n_keep_out, void* workspace, int64_t workspace_bytes, void* stream);

/* Start of a data-parallel Mapper.mapping call: certainty0_out <- certainty (device copy on `stream`). */
int pin_dp_cert_snapshot(const float* certainty, float* certainty0_out, int32_t n, void* stream);
/* The two element-wise halves of the certainty merge, usable with any transport:
 * delta_out = certainty - certainty0;  certainty = certainty0 + delta_sum. */
int pin_dp_cert_delta(const float* certainty, const float* certainty0, float* delta_out, int32_t n, void* stream);
int pin_dp_cert_apply(float* certainty, const float* certainty0, const float* delta_sum, int32_t n, void* stream);

/* End of a data-parallel Mapper.mapping call: certainty <- certainty0 + SUM_ranks(certainty - certainty0)
 * (certainty0 = the values before the call) and ts_update <- MAX_ranks(ts_update), the two all-reduces fused in
 * one RCCL group.  scratch [n] floats. */
int pin_dp_sync_side_effects(void* comm, float* certainty, const float* certainty0, float* scratch,
                             int32_t* ts_update, int32_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIN_ABI_H */
