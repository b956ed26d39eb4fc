"""Minimal stand-alone configuration carrying the attributes the hot path reads from the
reference's utils/config.py::Config (same names, same defaults; cited lines are config.py).
In drop-in mode the reference's own Config object is passed instead -- the classes only use
attribute access."""
from __future__ import annotations

import torch


class PinConfig:
    def __init__(self, **over):
        self.device = "cuda"
        self.dtype = torch.float32
        self.silence = True
        # preprocessing (config.py:53-68, 216)
        self.deskew = False
        self.min_range = 2.5
        self.min_z = -5.0
        self.max_z = 80.0
        self.rand_downsample = False
        self.rand_down_r = 1.0
        self.vox_down_m = 0.05
        self.source_vox_down_m = 0.8
        self.adaptive_range_on = False
        self.kitti_correction_on = False
        self.correction_deg = 0.0
        # neural points (config.py:91-103)
        self.voxel_size_m = 0.3
        self.weighted_first = True
        self.layer_norm_on = False
        self.num_nei_cells = 2
        self.query_nn_k = 6
        self.use_mid_ts = False
        self.search_alpha = 0.2
        self.buffer_size = int(5e7)
        self.feature_dim = 8
        self.feature_std = 0.0
        self.color_on = False
        self.color_channel = 0
        self.semantic_on = False
        # semantic head (config.py:70-74, 143-144, 186)
        self.sem_class_count = 20
        self.sem_label_decimation = 1
        self.freespace_label_on = False
        self.sem_mlp_level = 1
        self.sem_mlp_hidden_dim = 64
        self.weight_s = 1.0
        # local map (config.py:113-116)
        self.diff_ts_local = 400.0
        self.local_map_travel_dist_ratio = 5.0
        self.local_map_radius = 50.0
        self.max_range = 60.0
        # decoder (config.py:138-157)
        self.mlp_bias_on = True
        self.mlp_leaky_relu = False
        self.geo_mlp_level = 1
        self.geo_mlp_hidden_dim = 64
        self.color_mlp_level = 1
        self.color_mlp_hidden_dim = 64
        self.use_gaussian_pe = False
        self.pos_encoding_band = 0
        self.pos_input_dim = 3
        self.main_loss_type = "bce"
        self.sigma_sigmoid_m = 0.1
        self.logistic_gaussian_ratio = 0.55
        self.surface_sample_range_m = 0.25
        # sampler / pool (config.py:124-136, 169-171)
        self.surface_sample_n = 3
        self.free_sample_begin_ratio = 0.3
        self.free_sample_end_dist_m = 1.0
        self.free_front_n = 2
        self.free_behind_n = 1
        self.dist_weight_on = True
        self.dist_weight_scale = 0.8
        self.behind_dropoff_on = False
        self.window_radius = 50.0
        self.pool_capacity = int(1e7)
        self.new_certainty_thre = 1.0
        self.pool_filter_freq = 10
        self.from_sample_points = True
        self.from_all_samples = False
        self.map_surface_ratio = 0.5
        self.prune_map_on = False
        self.max_prune_certainty = 3.0
        self.prune_freq_frame = 100
        self.adaptive_iters = False
        self.new_sample_ratio_less = 0.02
        self.new_sample_ratio_more = 0.15
        self.new_sample_ratio_restart = 0.3
        self.freeze_after_frame = 40
        # mapping (config.py:160-199)
        self.loss_weight_on = False
        self.numerical_grad = True
        self.gradient_decimation = 10
        self.num_grad_step_ratio = 0.2
        self.ekional_loss_on = True
        self.ekional_add_to = "all"
        self.weight_e = 0.5
        self.weight_i = 1.0
        self.proj_correction_on = False
        self.consistency_loss_on = False
        self.iters = 12
        self.opt_adam = True
        self.bs = 16384
        self.bs_new_sample = 2048
        self.lr = 0.01
        self.weight_decay = 0.0
        self.adam_eps = 1e-15
        self.pgo_on = False
        self.track_on = True
        self.wandb_vis_on = False
        # tracking (config.py:209-236)
        self.photometric_loss_on = False
        self.photometric_loss_weight = 0.01
        self.consist_wieght_on = True  # (sic) config.py:215
        self.reg_min_grad_norm = 0.5
        self.reg_max_grad_norm = 2.0
        self.max_sdf_ratio = 5.0
        self.max_sdf_std_ratio = 1.0
        self.reg_dist_div_grad_norm = False
        self.reg_GM_dist_m = 0.3
        self.reg_GM_grad = 0.1
        self.reg_lm_lambda = 1e-4
        self.reg_iter_n = 50
        self.reg_term_thre_deg = 0.01
        self.reg_term_thre_m = 0.001
        self.eigenvalue_check = True
        self.eigenvalue_ratio_thre = 0.005
        self.final_residual_ratio_thre = 0.6
        for k, v in over.items():
            if not hasattr(self, k):
                raise AttributeError(f"PinConfig has no attribute {k}")
            setattr(self, k, v)
        self.track_mask_query_nn_k = over.get("track_mask_query_nn_k", self.query_nn_k)
        # derived (config.py:556-562)
        self.infer_bs = self.bs * 32
