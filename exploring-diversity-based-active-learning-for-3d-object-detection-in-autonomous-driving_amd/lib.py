"""ctypes binding of libal3d_hip.so (the C ABI in include/al3d.h).

There is deliberately no fallback: if the shared library is missing or a symbol
is absent the import raises, and every entry point raises ``Al3dError`` on a
non-zero status.  Build with ``python __graft_entry__.py`` (or ``make -C csrc``).
"""
import ctypes
import os

# Load PyTorch's bundled HIP runtime first: libal3d_hip.so needs libamdhip64.so.7 by
# SONAME, and one process must not end up with two HIP runtimes.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libal3d_hip.so")

c_i64, c_int, c_dbl, c_flt, c_p = (ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                   ctypes.c_float, ctypes.c_void_p)


class Al3dError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/al3d.h one for one
SIGNATURES = {
    "al3d_abi_version": (c_int, []),
    "al3d_last_error": (ctypes.c_char_p, []),
    "al3d_l1_distance_f32": (c_int, [c_p, c_i64, c_i64, c_int, c_p, c_p]),
    "al3d_l1_distance_rows_f32": (c_int, [c_p, c_i64, c_i64, c_int, c_i64, c_i64, c_p, c_p]),
    "al3d_combine_maps_f64": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_int, c_dbl, c_dbl, c_dbl,
                                      c_dbl, c_p, c_p]),
    "al3d_euclid_map_f64": (c_int, [c_p, c_p, c_i64, c_p, c_p]),
    "al3d_max_finite_f64": (c_int, [c_p, c_i64, c_p, c_p]),
    "al3d_greedy_workspace_bytes": (c_i64, [c_i64, c_int]),
    "al3d_greedy_kcenter_f64": (c_int, [c_p, c_p, c_i64, c_p, c_i64, c_i64, c_p, c_dbl, c_dbl,
                                        c_dbl, c_int, c_p, c_i64, c_p, c_p, c_p]),
    "al3d_greedy_kcenter_f32": (c_int, [c_p, c_p, c_i64, c_p, c_i64, c_i64, c_p, c_dbl, c_dbl,
                                        c_dbl, c_int, c_p, c_i64, c_p, c_p, c_p]),
    "al3d_frame_entropy_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_p, c_p]),
    "al3d_frame_weighted_entropy_f32": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_p, c_int, c_p, c_p]),
    "al3d_mask_map_f32": (c_int, [c_p, c_i64, c_p, c_p]),
    "al3d_scale_rows_f32": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_p, c_p]),
    "al3d_minmax_norm_f32": (c_int, [c_p, c_i64, c_p, c_p]),
    "al3d_argsort_workspace_bytes": (c_i64, [c_i64]),
    "al3d_argsort_desc_f32": (c_int, [c_p, c_i64, c_p, c_p, c_p]),
    "al3d_knn_2d_f64": (c_int, [c_p, c_i64, c_int, c_p, c_p, c_p]),
    "al3d_apsp_workspace_bytes": (c_i64, [c_i64, c_int]),
    "al3d_apsp_knn_f64": (c_int, [c_p, c_p, c_i64, c_int, c_p, c_p, c_p]),
    "al3d_apsp_knn_rows_f64": (c_int, [c_p, c_p, c_i64, c_int, c_i64, c_i64, c_p, c_p, c_p]),
    "al3d_merge_sweeps_workspace_bytes": (c_i64, [c_i64]),
    "al3d_merge_sweeps_f32": (c_int, [c_p, c_p, c_int, c_i64, c_p, c_p, c_p, ctypes.c_float, c_p, c_p, c_p, c_p]),
    "al3d_voxelize_grid_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "al3d_voxelize_grid_init": (c_int, [c_p, c_int, c_int, c_int, c_int, c_p]),
    "al3d_voxelize_workspace_bytes": (c_i64, [c_i64, c_int, c_int]),
    "al3d_voxelize_mean_f32": (c_int, [c_p, c_p, c_i64, c_int, c_int, c_p, c_p, c_p, c_int, c_int,
                                       c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "al3d_sp_fill_i32": (c_int, [c_p, c_i64, c_int, c_p]),
    "al3d_sp_scatter_index": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_int, c_p]),
    "al3d_sp_subm_table": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_int, c_int, c_int,
                                   c_p, c_p]),
    "al3d_sp_down_claim": (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_p,
                                   c_p, c_int, c_p]),
    "al3d_sp_down_sites_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "al3d_sp_down_sites": (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_p,
                                   c_p, c_int, c_p, c_p]),
    "al3d_sp_mask_window_sort_workspace_bytes": (c_i64, [c_int]),
    "al3d_sp_mask_window_sort": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_int, c_p, c_p, c_p]),
    "al3d_sp_down_sites_blocked_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "al3d_sp_down_sites_blocked": (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_p,
                                           c_p, c_int, c_p, c_p]),
    "al3d_sp_block_shape": (c_int, [c_int, c_int, c_p, c_p]),
    "al3d_sp_block_plan": (c_int, [c_p, c_i64, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p]),
    "al3d_sp_conv_blk_f16x3": (c_int, [c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p,
                                       c_int, c_p, c_int, c_int, c_p]),
    "al3d_sp_down_table": (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_p,
                                   c_p]),
    "al3d_sp_conv_f32": (c_int, [c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int, c_p, c_int,
                                 c_p]),
    "al3d_sp_conv_mfma_f32": (c_int, [c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int, c_p,
                                      c_int, c_p]),
    "al3d_sp_conv_bf16x6": (c_int, [c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int, c_p,
                                           c_int, c_p]),
    "al3d_sp_conv_wave_bf16x6": (c_int, [c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int, c_p,
                                                c_int, c_p]),
    "al3d_sp_conv_wave2_bf16x6": (c_int, [c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int, c_p,
                                                 c_int, c_p]),
    "al3d_sp_conv_wave2_f16x3": (c_int, [c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int, c_p,
                                                 c_int, c_p]),
    "al3d_sp_pack_glds_f16x3_elems": (c_i64, [c_int, c_int, c_int]),
    "al3d_sp_pack_glds_f16x3": (c_int, [c_p, c_int, c_int, c_int, c_p, c_p]),
    "al3d_sp_conv_glds_f16x3": (c_int, [c_p, c_p, c_int, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int,
                                        c_p, c_int, c_p]),
    "al3d_sp_tile_ranges": (c_int, [c_p, c_i64, c_int, c_int, c_p, c_p]),
    "al3d_sp_conv_rng_f16x3": (c_int, [c_p, c_p, c_int, c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int,
                                       c_p, c_int, c_int, c_p]),
    "al3d_sp_raster_perm_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "al3d_sp_raster_perm": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p]),
    "al3d_sp_rows_gather_pad_f32": (c_int, [c_p, c_p, c_i64, c_int, c_int, c_int, c_p, c_p]),
    "al3d_sp_tile_items_workspace_bytes": (c_i64, [c_int]),
    "al3d_sp_tile_items": (c_int, [c_p, c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p]),
    "al3d_sp_pack_r16_f16x3_elems": (c_i64, [c_int]),
    "al3d_sp_pack_r16_f16x3": (c_int, [c_p, c_int, c_p, c_p]),
    "al3d_sp_conv_r16_f16x3": (c_int, [c_p, c_p, c_int, c_p, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int,
                                       c_p, c_int, c_int, c_int, c_p]),
    "al3d_sp_rows_convert_f16x3": (c_int, [c_p, c_i64, c_int, c_int, c_p, c_p]),
    "al3d_sp_conv_glds_f16x3_io": (c_int, [c_p, c_p, c_int, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int,
                                           c_p, c_int, c_int, c_p]),
    "al3d_sp_conv_wave2_f16x3_tiles_io": (c_int, [c_p, c_p, c_int, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int,
                                                  c_p, c_int, c_int, c_p]),
    "al3d_sp_conv_wave2_f16x3_tiles": (c_int, [c_p, c_p, c_int, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p, c_int,
                                               c_p, c_int, c_p]),
    "al3d_sp_table_pitch": (c_int, [c_int]),
    "al3d_sp_subm_table_tiles": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_int, c_int, c_int,
                                         c_p, c_int, c_p, c_p]),
    "al3d_sp_down_table_tiles": (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_p,
                                         c_int, c_p, c_p]),
    "al3d_merge_bf16x3": (c_int, [c_p, c_i64, c_p, c_p]),
    "al3d_sp_to_dense_nhwc": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p]),
    "al3d_head_decode_nms": (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                     c_flt, c_flt, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "al3d_head_decode_nms_workspace_bytes": (c_i64, [c_int, c_int, c_p]),
    "al3d_box_decode_f32": (c_int, [c_p, c_p, c_i64, c_p, c_p]),
    "al3d_vfe_mean_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_p, c_p]),
    "al3d_conv2d_nhwc_f32": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 11 + [c_p]),
    "al3d_deconv2x2_nhwc_f32": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p]),
    "al3d_split_bf16x3": (c_int, [c_p, c_i64, c_p, c_p]),
    "al3d_conv2d_nhwc_bf16x6": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 11 + [c_p]),
    "al3d_deconv2x2_nhwc_bf16x6": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p]),
    "al3d_split_f16x3": (c_int, [c_p, c_i64, c_int, c_p, c_p]),
    "al3d_conv2d_nhwc_f16x3": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 11 + [c_p]),
    "al3d_deconv2x2_nhwc_f16x3": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p]),
    "al3d_pack_f16x3_frag": (c_int, [c_p, c_int, c_int, c_p, c_p]),
    "al3d_pack_f16x3_frag16": (c_int, [c_p, c_int, c_int, c_p, c_p]),
    "al3d_conv3x3_nhwc_f16x3_frag16": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p]),
    "al3d_conv3x3_nhwc_f16x3_frag": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p]),
    "al3d_conv3x3_nhwc_f16x3_frag_io": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 9 + [c_p]),
    "al3d_pack_f16x3_bstream_elems": (c_i64, [c_int, c_int, c_int]),
    "al3d_pack_f16x3_bstream": (c_int, [c_p, c_int, c_int, c_int, c_p, c_p]),
    "al3d_conv2d_nhwc_f16x3_bstream": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 11 + [c_p]),
    "al3d_deconv2x2_nhwc_f16x3_bstream": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p]),
    "al3d_pack_f16x3_dma": (c_int, [c_p, c_int, c_int, c_int, c_p, c_p]),
    "al3d_conv2d_nhwc_f16x3_dma": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 11 + [c_p, c_int, c_int, c_p]),
    "al3d_deconv2x2_nhwc_f16x3_dma": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p, c_int, c_int, c_p]),
    "al3d_stream_create_cu_mask": (c_int, [c_int, c_int, c_p]),
    "al3d_lss_geometry_workspace_bytes": (c_i64, [c_int]),
    "al3d_lss_geometry_f32": (c_int, [c_p, c_i64, c_p, c_int, c_p, c_p, c_p]),
    "al3d_lss_depth_softmax_f32": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_p]),
    "al3d_lss_upsample_cat_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p]),
    "al3d_jpeg_header": (c_int, [c_p, c_i64, c_p, c_p]),
    "al3d_jpeg_entropy_decode": (c_int, [c_p, c_i64, c_p, c_i64]),
    "al3d_jpeg_workspace_bytes": (c_i64, [c_p, c_int]),
    "al3d_jpeg_idct_rgb_u8": (c_int, [c_p, c_p, c_p, c_int, c_p, c_p, c_p]),
    "al3d_cat2_nhwc_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p]),
    "al3d_lss_upsample_cat_mode_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p]),
    "al3d_lss_dtransform01_f32": (c_int, [c_p, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_p]),
    "al3d_lss_depth_image_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "al3d_lss_depth_image_f32": (c_int, [c_p, c_i64, c_int, c_p, c_int, c_p, c_int, c_int, c_p, c_p, c_p]),
    "al3d_bev_pool_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "al3d_bev_pool_f32": (c_int, [c_p, c_p, c_i64, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_p]),
    "al3d_bev_pool_lss_f32": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p,
                                      c_p, c_p]),
    "al3d_bev_pool_plan": (c_int, [c_p, c_i64, c_int, c_p, c_p, c_p, c_p, c_p]),
    "al3d_bev_pool_lss_apply_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p]),
    "al3d_reader_create": (c_int, [c_int, c_p]),
    "al3d_reader_destroy": (None, [c_p]),
    "al3d_reader_plan": (c_i64, [c_p, c_int, c_p]),
    "al3d_reader_submit": (c_int, [c_p, c_p, c_int, c_p, c_p, c_p, c_i64]),
    "al3d_reader_wait": (c_int, [c_p, c_int]),
    "al3d_merge_sweeps_batch_f32": (c_int, [c_p, c_p, c_int, c_i64, c_p, c_p, c_p, c_p, c_p, c_int, ctypes.c_float,
                                            c_p, c_p, c_p, c_p]),
    "al3d_merge_sweeps_batch_rule_f32": (c_int, [c_p, c_p, c_int, c_i64, c_p, c_p, c_p, c_p, c_p, c_int, ctypes.c_float,
                                                 c_int, c_p, c_p, c_p, c_p]),
    "al3d_merge_sweeps_batch_range_f32": (c_int, [c_p, c_p, c_int, c_i64, c_p, c_p, c_p, c_p, c_p, c_int, ctypes.c_float,
                                                  c_int, c_p, c_p, c_p, c_p, c_p]),
    "al3d_tf_proposals_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "al3d_tf_proposals_f32": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint, c_int, c_p, c_int, c_p, c_p, c_p,
                                      c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "al3d_image_resample_ksize": (c_int, [c_int, c_int, c_int]),
    "al3d_image_resample_coeffs": (c_int, [c_int, c_int, c_int, c_p, c_p]),
    "al3d_image_aug_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "al3d_image_aug_normalize_u8": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p,
                                            c_int, c_p, c_p, c_int, c_p, c_p, c_int, c_int, c_p, c_p, c_p, c_p]),
    "al3d_gap_parts_count": (c_int, [c_int, c_int, c_int]),
    "al3d_conv2d_nhwc_f16x3_gap": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 11 + [c_p, c_int, c_p]),
    "al3d_deconv2x2_nhwc_f16x3_gap": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 8 + [c_p, c_int, c_p]),
    "al3d_gap_reduce_parts_f32": (c_int, [c_p, c_int, c_int, c_int, c_i64, c_p, c_p]),
    "al3d_gap_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "al3d_gap_nhwc_f32": (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_p, c_p]),
    "al3d_pack_f16x3_wino": (c_int, [c_p, c_int, c_int, c_p, c_p]),
    "al3d_conv3x3_nhwc_f16x3_wino": (c_int, [c_p, c_p, c_p, c_p, c_p] + [c_int] * 9 + [c_p]),
    "al3d_tok_patch_rows_f32": (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_p]),
    "al3d_tok_layernorm_f32": (c_int, [c_p, c_p, c_i64, c_int, c_int, c_int, c_p, c_p, c_flt, c_int, c_p, c_p]),
    "al3d_tok_patch_embed_image_bytes": (c_i64, []),
    "al3d_tok_patch_embed_f16x3": (c_int, [c_p, c_int, c_int, c_int, c_p, c_flt, c_p, c_p, c_p, c_flt, c_p, c_p]),
    "al3d_tok_attn_block_image_bytes": (c_i64, [c_int]),
    "al3d_tok_attn_block_f16x3": (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_flt, c_p, c_flt, c_p, c_flt,
                                          c_p, c_p, c_flt, c_p]),
    "al3d_tok_mlp_image_bytes": (c_i64, [c_int, c_int]),
    "al3d_tok_mlp_f16x3": (c_int, [c_p, c_i64, c_int, c_int, c_p, c_p, c_flt, c_p, c_flt, c_p, c_flt, c_p, c_p]),
    "al3d_tok_linear_f16x3": (c_int, [c_p, c_int, c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_p, c_int, c_p, c_p,
                                      c_int, c_int, c_p]),
    "al3d_tok_mha16_workspace_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "al3d_tok_mha16_f32": (c_int, [c_p, c_int, c_p, c_int, c_p, c_int, c_int, c_int, c_int, c_int, c_flt, c_p, c_int, c_p,
                                   c_p]),
    "al3d_tok_window_attention_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_flt, c_int, c_p,
                                              c_p]),
    "al3d_tok_window_attention_tokens_f32": (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_flt, c_int, c_p,
                                                     c_p]),
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol (no compute)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Al3dError(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Run `python __graft_entry__.py` (build()) first; there is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().al3d_last_error().decode("utf-8", "replace")
        raise Al3dError(f"{what} failed with status {status}: {msg}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)
