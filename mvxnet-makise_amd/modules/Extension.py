"""Loader of the HIP C-ABI library -- the drop-in boundary.

Stands where the reference's ``modules/Extension.py:1-3`` JIT-built the pybind11
module ``cpp/voxelutil.cpp``: importing this module gives ``lib`` (ctypes handle
of ``libmvx_hip.so``, prototypes from ``include/mvx_hip.h``) and ``cpp``, an
object with the reference extension's ``_group`` signature
(``cpp/voxelutil.cpp:325,365``) implemented on the GPU.

There is no CPU fallback: a missing library or a call without a GPU raises.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MVX_HIP_LIB', os.path.join(os.path.dirname(_HERE), 'lib', 'libmvx_hip.so'))
ABI_VERSION = 8

_p = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f64 = ctypes.c_double
_f32 = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/mvx_hip.h one to one
PROTOTYPES = {
    'mvx_abi_version': (_i32, []),
    'mvx_tuning_set': (_i32, [_i32, _i64]),
    'mvx_launch_count': (ctypes.c_uint64, []),
    'mvx_voxelize_workspace_bytes': (_sz, [_i32, _i32]),
    'mvx_voxelize': (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _f64, _f64, _f64, _f64, _f64, _f64,
                            _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mvx_voxelize_frames': (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _f64, _f64, _f64, _f64, _f64, _f64,
                                   _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mvx_crop_workspace_bytes': (_sz, [_i32, _i32]),
    'mvx_crop_points': (_i32, [_p, _p, _i32, _i32, _i32, _p, _i32, _p, _p, _f64, _f64, _i32, _p, _p, _p, _p, _sz, _p]),
    'mvx_crop_project_workspace_bytes': (_sz, [_i32, _i32]),
    'mvx_crop_project_points': (_i32, [_p, _p, _i32, _i32, _i32, _p, _p, _p, _f64, _f64, _p, _p, _p, _i32, _p, _p, _sz, _p]),
    'mvx_lidar2img': (_i32, [_p, _i32, _i64, _p, _p, _i32, _p, _i32, _i32, _i32, _p, _p]),
    'mvx_scatter_voxels': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _i32, _i32, _p, _p]),
    'mvx_gather_voxels': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_cl_to_bev': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_index_grid_bytes': (_sz, [_i32, _i32, _i32]),
    'mvx_index_grid': (_i32, [_p, _i32, _i32, _i32, _i32, _p, _p, _p]),
    'mvx_sparse_conv_output': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_sparse_conv_gather_dz': (_i32, [_p, _p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_row_stats': (_i32, [_p, _p, _i64, _i32, _p]),
    'mvx_bn_finalize': (_i32, [_p, _f64, _f64, _p, _i32, _p]),
    'mvx_bn_apply': (_i32, [_p, _p, _p, _i64, _i32, _p]),
    'mvx_bn_backward_scratch_bytes': (_sz, [_i32]),
    'mvx_bn_relu_backward': (_i32, [_p, _p, _p, _f64, _p, _p, _p, _p, _i64, _i32, _i32, _p]),
    'mvx_linear_splitk_workspace_bytes': (_sz, [_i64, _i32]),
    'mvx_linear_forward': (_i32, [_p, _i32, _p, _i32, _i32, _p, _p, _i32, _p, _p, _i64, _i32, _i32, _i32, _p, _sz, _p]),
    'mvx_linear_wgrad_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'mvx_linear_wgrad': (_i32, [_p, _i32, _p, _i32, _p, _i64, _i32, _i32, _i32, _p, _sz, _p]),
    'mvx_split_planes_bytes': (_sz, [_i64, _i32, _i32]),
    'mvx_split_rows': (_i32, [_p, _i32, _i64, _i32, _p, _i32, _f32, _p]),
    'mvx_linear_forward_pre_frames': (_i32, [_p, _p, _p, _p, _i32, _p, _p, _i64, _i32, _i32, _i32, _f32, _p, _f64, _p, _p, _i32, _p]),
    'mvx_linear_wgrad_pre_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'mvx_linear_wgrad_pre': (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _f32, _p, _sz, _p]),
    'mvx_linear_wgrad_pre_rows': (_i32, [_p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _p, _sz, _p]),
    'mvx_vfe_bn_max_concat': (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _p]),
    'mvx_vfe_max_concat_backward': (_i32, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _p]),
    'mvx_bn_segment_max': (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _p]),
    'mvx_segment_max_backward': (_i32, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _p]),
    'mvx_voxel_row_offsets': (_i32, [_p, _i32, _i32, _i32, _p, _p, _p, _p]),
    'mvx_vfe_compact_input': (_i32, [_p, _i32, _p, _p, _i32, _i32, _i32, _p, _p]),
    'mvx_vfe_compact_input_backward': (_i32, [_p, _i32, _i32, _i32, _p, _p, _p]),
    'mvx_row_compact_workspace_bytes': (_sz, [_i64]),
    'mvx_row_compact_map': (_i32, [_p, _i32, _i64, _p, _p, _p, _p, _sz, _p]),
    'mvx_feature_sample': (_i32, [_p, _i32, _i64, _p, _p, _p, _i32, _i32, _f32, _f32, _f32, _p, _p, _p]),
    'mvx_feature_sample_rows': (_i32, [_p, _i32, _p, _i32, _p, _p, _i32, _i32, _f32, _f32, _f32, _p, _p, _p]),
    'mvx_expand_rows': (_i32, [_p, _p, _i32, _p, _i64, _i32, _p]),
    'mvx_expand_rows_backward': (_i32, [_p, _p, _i32, _p, _p, _i64, _i32, _p]),
    'mvx_conv3d_packed_weight_bytes': (_sz, [_i32, _i32]),
    'mvx_conv3d_pack_weights': (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    'mvx_conv3d_tile_shape': (None, [_p, _p]),
    'mvx_conv3d_forward': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_conv3d_packed_weight_bytes_split': (_sz, [_i32, _i32, _i32]),
    'mvx_conv3d_pack_weights_split': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _p]),
    'mvx_conv3d_forward_split': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_conv3d_dgrad_split': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_conv3d_wgrad_split': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    'mvx_conv3d_dgrad': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_conv3d_wgrad_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'mvx_conv3d_wgrad': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    'mvx_activity_dilate': (_i32, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    'mvx_conv3d_background': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_bn_background': (_i32, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    'mvx_conv3d_forward_bg': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p,
                                     _i32, _p, _p, _f64, _f64, _p, _p, _p]),
    'mvx_linear_forward_bn': (_i32, [_p, _i32, _p, _i32, _i32, _p, _p, _i32, _p, _p, _i64, _i32, _i32, _i32, _p, _f64, _f64, _p, _p]),
    'mvx_conv3d_wgrad_bg_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'mvx_conv3d_wgrad_bg': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    'mvx_conv3d_forward_bg_split': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p,
                                           _i32, _p]),
    'mvx_conv3d_dgrad_tiles_split': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_conv3d_wgrad_bg_split_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'mvx_conv3d_wgrad_bg_split': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    'mvx_conv3d_forward_bg_split_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p,
                                                  _p, _i32, _p, _i32, _p]),
    'mvx_conv3d_dgrad_tiles_split_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _i32, _p]),
    'mvx_conv3d_wgrad_bg_split_workspace_bytes_frames': (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    'mvx_conv3d_wgrad_bg_split_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _sz,
                                                _i32, _p]),
    'mvx_conv2d_forward_split_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_conv2d_dgrad_split_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_conv2d_wgrad_split_workspace_bytes_frames': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'mvx_conv2d_wgrad_split_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _sz, _i32, _p]),
    'mvx_plane_tap_sums_workspace_bytes': (_sz, [_i32, _i32]),
    'mvx_plane_tap_sums': (_i32, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _p]),
    'mvx_tile_dilate_flags': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_conv3d_input_grad_sums': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_conv3d_dgrad_tiles': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    'mvx_bn_relu_backward_tiles_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'mvx_bn_relu_backward_tiles': (_i32, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p, _sz, _p]),
    # frame-set forms (one launch for all frames of a step; include/mvx_hip.h "Frame sets")
    'mvx_linear_forward_bn_frames': (_i32, [_p, _i32, _p, _i32, _i32, _p, _p, _i32, _p, _p, _i64, _i32, _i32, _i32, _p, _f64, _p, _p, _i32, _p]),
    'mvx_bn_finalize_frames': (_i32, [_p, _f64, _f64, _p, _i32, _i32, _p]),
    'mvx_bn_apply_frames': (_i32, [_p, _p, _p, _i64, _i32, _p, _i32, _p]),
    'mvx_bn_backward_scratch_bytes_frames': (_sz, [_i32, _i32]),
    'mvx_split_operand_amax': (_i32, [_p, _p]),
    'mvx_tensor_amax': (_i32, [_p, _i64, _p, _i32, _p]),
    'mvx_split_f16_weight_check': (_i32, [_p, _i64, _p, _p]),
    'mvx_bn_relu_backward_frames': (_i32, [_p, _p, _p, _f64, _p, _p, _p, _p, _i64, _i32, _i32, _p, _i32, _p, _p]),
    'mvx_bn_relu_backward_planes_frames': (_i32, [_p, _p, _p, _f64, _p, _p, _p, _p, _i64, _i32, _i32, _p, _i32, _p, _p]),
    'mvx_bn_relu_backward_planes_part_frames': (_i32, [_p, _p, _p, _f64, _p, _p, _p, _p, _i64, _i32, _i32, _p, _i32, _i32, _i32, _p,
                                                       _p]),
    'mvx_vfe_bn_max_concat_frames': (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _p, _p]),
    'mvx_bn_segment_max_frames': (_i32, [_p, _p, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _p, _p]),
    'mvx_voxel_row_offsets_frames': (_i32, [_p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    'mvx_vfe_compact_input_frames': (_i32, [_p, _i32, _p, _p, _i32, _i32, _i32, _p, _p, _p]),
    'mvx_vfe_compact_input_pitch_frames': (_i32, [_p, _i32, _p, _p, _i32, _i32, _i32, _p, _i32, _p, _p]),
    'mvx_vfe_compact_input_backward_frames': (_i32, [_p, _i32, _i32, _i32, _p, _p, _p, _p]),
    'mvx_row_compact_map_frames': (_i32, [_p, _i32, _i64, _p, _p, _p, _p, _sz, _p, _p, _p]),
    'mvx_feature_sample_rows_frames': (_i32, [_p, _i32, _p, _i32, _p, _p, _i32, _i32, _f32, _f32, _f32, _p, _p, _p, _p, _p]),
    'mvx_feature_sample_rows_planes_frames': (_i32, [_p, _i32, _p, _i32, _p, _p, _i32, _i32, _f32, _f32, _f32, _p, _p, _p, _p, _p,
                                                     _i64, _p]),
    'mvx_index_grid_bytes_frames': (_sz, [_i32, _i32, _i32, _i32]),
    'mvx_index_grid_frames': (_i32, [_p, _i32, _i32, _i32, _i32, _p, _p, _p, _p]),
    'mvx_sparse_conv_output_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_sparse_conv_output_tiles_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_sparse_conv_gather_dz_frames': (_i32, [_p, _p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'mvx_activity_dilate_frames': (_i32, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p]),
    'mvx_tile_dilate_flags_frames': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    'mvx_conv3d_background_frames': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    'mvx_bn_apply_tiles_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_bn_apply_tiles_bev_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_bn_apply_tiles_read_frames': (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_tile_read_flags_frames': (_i32, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    'mvx_conv3d_background_taps_frames': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    'mvx_bn_background_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _p, _p, _i32, _p]),
    'mvx_conv3d_forward_bg_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p,
                                            _i32, _p, _p, _f64, _f64, _p, _p, _i32, _p]),
    'mvx_conv3d_dgrad_tiles_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p]),
    'mvx_conv3d_input_grad_sums_frames': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    'mvx_conv3d_wgrad_bg_workspace_bytes_frames': (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    'mvx_conv3d_wgrad_bg_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _sz, _i32, _p]),
    'mvx_bn_relu_backward_tiles_workspace_bytes_frames': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'mvx_bn_relu_backward_tiles_frames': (_i32, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _sz, _i32, _p]),
    'mvx_cl_to_bev_frames': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_bev_to_cl_broadcast': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_conv2d_forward_frames': (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _f64, _p, _p, _i32, _p]),
    'mvx_conv2d_dgrad_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    'mvx_conv2d_wgrad_workspace_bytes_frames': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'mvx_conv2d_wgrad_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _sz, _i32, _p]),
    'mvx_space_to_depth_frames': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_d2s_bn_apply_frames': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_bn_apply_strided_frames': (_i32, [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p]),
    'mvx_row_stats_frames': (_i32, [_p, _p, _i64, _i32, _i32, _p]),
    'mvx_bbox_pairwise': (_i32, [_p, _i32, _p, _i32, _i32, _p, _p]),
    'mvx_classify_anchors_workspace_bytes': (_sz, [_i32, _i32, _i32]),
    'mvx_classify_anchors': (_i32, [_p, _i32, _p, _i32, _i32, _i32, _p, _p, _f32, _f32, _i32, _p, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    'mvx_classify_anchors_frames': (_i32, [_p, _p, _i32, _p, _i32, _i32, _i32, _p, _p, _f32, _f32, _i32, _p, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    'mvx_voxel_loss': (_i32, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _i32, _i32, _p, _i32, _p,
                              _i32, _i32, _i32, _f32, _f32, _f32, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _p, _p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libmvx_hip.so not found at %s -- build it with `python __graft_entry__.py` '
            '(or `make -C mvxnet-makise_amd/csrc`). There is no CPU fallback.' % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(handle, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = handle.mvx_abi_version()
    if got != ABI_VERSION:
        raise ImportError('libmvx_hip ABI %d != expected %d: rebuild the library' % (got, ABI_VERSION))
    return handle


lib = _load()


class MvxHipError(RuntimeError):
    pass


MAX_FRAMES = 16       # MVX_MAX_FRAMES
ROWS_SINGLE, ROWS_FUSION, ROWS_VFE, ROWS_VOXELS, ROWS_GRID, ROWS_REAL = 0, 1, 2, 3, 4, 5      # MVX_ROWS_*


class FramesDesc(ctypes.Structure):
    """mvx_frames_t of include/mvx_hip.h: host-side descriptor of the frames that share a launch."""
    _fields_ = [('n_frames', ctypes.c_int32), ('t', ctypes.c_int32),
                ('real_off', ctypes.c_int32 * (MAX_FRAMES + 1)), ('vox_off', ctypes.c_int32 * (MAX_FRAMES + 1))]

    @classmethod
    def make(cls, vox_off, real_off, t):
        d = cls()
        d.n_frames = len(vox_off) - 1
        if not 1 <= d.n_frames <= MAX_FRAMES:
            raise MvxHipError('a frame set holds 1..%d frames' % MAX_FRAMES)
        d.t = int(t)
        for i, v in enumerate(vox_off):
            d.vox_off[i] = int(v)
        for i, v in enumerate(real_off):
            d.real_off[i] = int(v)
        return d

    def ref(self):
        return ctypes.byref(self)


def check(status, what):
    if status != 0:
        kind = 'argument error' if status < 0 else 'hipError_t'
        raise MvxHipError('%s failed: %s %d' % (what, kind, status))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MvxHipError('libmvx_hip needs device tensors; got a %s tensor (no CPU fallback)' % t.device)
    if not t.is_contiguous():
        raise MvxHipError('libmvx_hip needs contiguous tensors')
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def raw_stream(index=None):
    """Current HIP stream handle of a device as an int (no torch.cuda.Stream object is built)."""
    return _raw_stream(_cur_device() if index is None else index)


def stream():
    return ctypes.c_void_p(raw_stream())


def device():
    if not torch.cuda.is_available():
        raise MvxHipError('no GPU visible: the MVXNet hot path has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


class _Cpp:
    """The reference extension's python-visible surface (cpp/voxelutil.cpp:362-368): ``_group``,
    ``_classifyAnchors``, ``bboxOverlap``, ``bboxIntersection`` -- numpy in, numpy out like the pybind11
    module, computed by the HIP kernels."""

    @staticmethod
    def _boxes(b, dev):
        b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
        return torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(dev)

    @staticmethod
    def bboxOverlap(bboxes1, bboxes2):
        """IoU of every BEV box pair, (N,4,2) x (M,4,2) -> (N,M) f32 (cpp/voxelutil.cpp:96-116; see
        include/mvx_hip.h for the reference's corner-index bug that is not reproduced)."""
        from modules import _hip
        dev = device()
        return _hip.bbox_pairwise(_Cpp._boxes(bboxes1, dev), _Cpp._boxes(bboxes2, dev), True).cpu().numpy()

    @staticmethod
    def bboxIntersection(bboxes1, bboxes2):
        """Intersection area of every BEV box pair (cpp/voxelutil.cpp:118-136)."""
        from modules import _hip
        dev = device()
        return _hip.bbox_pairwise(_Cpp._boxes(bboxes1, dev), _Cpp._boxes(bboxes2, dev), False).cpu().numpy()

    @staticmethod
    def _classifyAnchors(gts, anchors, nls, nws, negThr, posThr):
        """``((px,py,pz), (nx,ny,nz), gi)`` int64 numpy arrays, the contract of cpp/voxelutil.cpp:138-316."""
        from modules import _hip
        from modules import Calc
        dev = device()
        g = _Cpp._boxes(gts, dev)
        a = _Cpp._boxes(anchors, dev)
        to_i64 = lambda v: torch.as_tensor(np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v, dtype=np.int64)).to(dev)
        radius = Calc._window_radius(g.cpu(), a[:2, :2].cpu()) if g.shape[0] else 2
        while True:
            pos, neg, gi, counts, status = _hip.classify_anchors(g, a, to_i64(nls), to_i64(nws), negThr, posThr, radius)
            n_pos, n_neg, st = counts.tolist() + status.tolist()
            if st & 1 and radius < 55:
                radius = min(55, 2 * radius)
                continue
            break
        if st:
            raise MvxHipError('_classifyAnchors: status %d (1 = window too small, 4 = centre outside the grid)' % st)
        pos, neg, gi = pos.cpu().numpy(), neg.cpu().numpy(), gi.cpu().numpy()
        return ((pos[0, :n_pos].copy(), pos[1, :n_pos].copy(), pos[2, :n_pos].copy()),
                (neg[0, :n_neg].copy(), neg[1, :n_neg].copy(), neg[2, :n_neg].copy()), gi[:n_pos].copy())

    @staticmethod
    def _group(pcd, idx, samplesPerVoxel):
        """``_group(pcd f32[P,>=4], idx i32[P,3], T) -> (voxel f32[V,T,7], (x,y,z) i64[V] x3,
        cnt i64[V])`` -- same contract as cpp/voxelutil.cpp:325-360 (numpy in, numpy out,
        inputs cast like pybind's forcecast)."""
        from modules import _hip
        pcd = np.ascontiguousarray(pcd, dtype=np.float32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        dev = device()
        P = pcd.shape[0]
        if P == 0:
            z = np.zeros(0, np.int64)
            return np.zeros((0, samplesPerVoxel, 7), np.float32), (z, z.copy(), z.copy()), z.copy()
        res = _hip.voxelize(torch.from_numpy(pcd).to(dev)[None], None, None,
                            (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), int(samplesPerVoxel), 7,
                            ext_idx=torch.from_numpy(idx).to(dev)[None])
        V = int(res.n_voxels[0])
        voxel = res.voxels[0, :V].cpu().numpy()
        c = res.coords[0, :V].cpu().numpy()
        # the native function leaves cols 3:6 zero; the centroid columns are added by the
        # python caller (Preprocessing.py:71-72)
        voxel[..., 3:6] = 0
        cnt = res.counts[0, :V].cpu().numpy().astype(np.int64)
        return voxel, (c[:, 1].copy(), c[:, 2].copy(), c[:, 3].copy()), cnt


cpp = _Cpp()
