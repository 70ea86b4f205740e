"""ctypes binding of libdgg_hip.so (include/dgg_hip.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("DGG_HIP_SO", os.path.join(_HERE, "libdgg_hip.so"))   # override: diagnostic builds only

_vp, _i64, _i32, _u32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_float, C.c_size_t

# name -> argtypes, exactly the prototypes of include/dgg_hip.h
PROTOTYPES = {
    "dgg_linear_fwd": [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "dgg_linear_bwd": [_vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_gemm_tn_acc": [_vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp],
    "dgg_gemm_tn_ws_floats": [_i64, _i32, _i32],
    "dgg_linear_pack_weights": [_i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "dgg_linear_fwd_multi": [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "dgg_gemm_tn_multi": [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp],
    "dgg_gemm_tn_multi_ws_floats": [_i64, _i32, _i32],
    "dgg_gemm_tn_pairs": [_i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "dgg_linear_bwd_ws_floats": [_i64, _i32, _i32],
    "dgg_degree_stats": [_vp, _i64, _vp, _vp, _vp],
    "dgg_degree_stats_ws_bytes": [],
    "dgg_knet_x_fwd": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_knet_x_fwd_mfma": [_vp, _i64, _i32] + [_vp] * 10 + [_vp],
    "dgg_knet_x_bwd_reg": [_vp, _i64, _i32] + [_vp] * 16 + [_i32, _vp, _vp],
    "dgg_knet_x_bwd_ws_bytes": [_i64, _i32],
    "dgg_knet_x_bwd_nodes": [_i64, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_knet_input_deg_fwd": [_vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "dgg_knet_feat": [_vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "dgg_knet_out_fwd": [_vp, _vp, _i64, _vp, _vp, _vp],
    "dgg_knet_out_bwd": [_vp, _vp, _vp, _i64, _vp, _vp],
    "dgg_knet_deg_fwd": [_vp, _i64, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "dgg_knet_deg_bwd_sums": [_vp, _i64, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp],
    "dgg_allpairs_topk": [_vp, _i64, _i32, _i64, _i64, _f32, _i32, _vp, _i64, _u32, _u32, _i32, _vp, _vp, _vp, _i32, _vp, _sz, _vp],
    "dgg_allpairs_workspace_bytes": [_i64, _i32, _i32, _i32],
    "dgg_allpairs_sweep_ctl_offset_bytes": [_i64, _i64, _i32],
    "dgg_allpairs_rsym_ctl_offset_bytes": [_i64, _i64],
    "dgg_allpairs_ranked_probe": [_vp, _i64, _i32, _i64, _i64, _f32, _u32, _u32, _vp, _i32, _i32, _vp, _vp, _vp],
    "dgg_allpairs_topk_ranked_softk_lp": [_vp, _i64, _i32, _i64, _i64, _f32, _u32, _u32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "dgg_allpairs_topk_ranked_softk": [_vp, _i64, _i32, _i64, _i64, _f32, _u32, _u32, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "dgg_literal_hard_fwd": [_vp, _i64, _i32, _vp, _vp, _f32, _i32, _vp, _i64, _u32, _u32, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp],
    "dgg_literal_hard_bwd": [_vp, _vp, _i64, _i32, _vp, _vp],
    "dgg_allpairs_topk_ranked_softk_dseed": [_vp, _i64, _i32, _i64, _i64, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "dgg_chunk_layout": [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp],
    "dgg_allpairs_anywide_ws_bytes": [_i64, _i64, _i64, _i32],
    "dgg_allpairs_rowmin_ws_bytes": [_i64, _i64, _i32],
    "dgg_allpairs_rowmin_bound": [_vp, _i64, _i32, _i64, _i64, _f32, _vp, _vp, _sz, _vp],
    "dgg_allpairs_topk_anywide": [_vp, _i64, _i32, _i64, _i64, _f32, _i32, _u32, _u32, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _sz, _vp],
    "dgg_allpairs_topk_ranked_wide": [_vp, _i64, _i32, _i64, _i64, _f32, _u32, _u32, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_partp_build_chunked": [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _vp],
    "dgg_ell_spmm_act_fwd_chunked": [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _vp, _vp],
    "dgg_ell_conv_bwd_partp_chunked": [_vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_softk_edge_bwd_partp_chunked": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _i32, _i32, _i32, _vp,
                                         _i64, _vp, _vp, _vp, _i32, _i32, _vp],
    "dgg_partp_gather_rec": [_vp, _i64, _i64, _vp, _vp, _vp],
    "dgg_edgelist_topk": [_vp, _i64, _i32, _vp, _vp, _f32, _i32, _vp, _i64, _u32, _u32, _i32, _vp, _vp, _vp],
    "dgg_edgelist_topk_softk": [_vp, _i64, _i32, _vp, _vp, _f32, _i32, _vp, _i64, _u32, _u32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_edge_mlp_fwd": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "dgg_edgelist_topk_p": [_vp, _i64, _vp, _vp, _i32, _vp, _i64, _u32, _u32, _i32, _vp, _vp, _vp, _vp],
    "dgg_edge_mlp_bwd": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "dgg_edge_mlp_bwd_partp": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    "dgg_csr_softk_fwd": [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _i64, _u32, _u32, _i32, _vp, _vp, _vp, _vp],
    "dgg_csr_softk_bwd": [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "dgg_csr_rank_ramp_fwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_csr_rank_ramp_bwd": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_csr_noisy_sigmoid_fwd": [_vp, _vp, _i64, _vp, _vp],
    "dgg_csr_noisy_sigmoid_bwd": [_vp, _vp, _i64, _vp, _vp],
    "dgg_csr_rank_cut_fwd": [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "dgg_csr_rank_cut_bwd": [_vp, _vp, _i64, _i32, _vp, _vp],
    "dgg_dense_rows_fwd": [_vp, _i32, _i64, _i32, _vp, _f32, _i32, _vp, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp],
    "dgg_dense_rows_bwd": [_vp, _i32, _i64, _i32, _vp, _f32, _i32, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_dense_pairs_dx": [_vp, _i32, _i64, _i32, _vp, _vp, _vp],
    "dgg_feat_softmax_fwd": [_vp, _i64, _i32, _vp, _vp],
    "dgg_feat_softmax_bwd": [_vp, _vp, _i64, _i32, _vp, _vp],
    "dgg_csr_uvdist_fwd": [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp],
    "dgg_csr_uvdist_bwd": [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp],
    "dgg_csr_row_sum": [_vp, _vp, _i64, _vp, _vp],
    "dgg_csr_normalize_fwd": [_vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "dgg_csr_norm_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "dgg_csr_spmm_fwd": [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "dgg_csr_bg_softmax_fwd": [_vp, _vp, _i64, _vp, _vp, _vp],
    "dgg_csr_bg_softmax_bwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "dgg_masked_dense_sum": [_vp, _i64, _i32, _f32, _u32, _u32, _i32, _vp, _vp],
    "dgg_pair_keep": [_vp, _vp, _i64, _f32, _u32, _u32, _vp, _vp],
    "dgg_csr_spmm_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "dgg_select_scores": [_vp, _i64, _i64, _i32, _vp, _vp, _vp],
    "dgg_softk_fwd": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp],
    "dgg_ell_normalize_fwd": [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp],
    "dgg_ell_spmm_fwd": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "dgg_ell_spmm_bwd": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp],
    "dgg_norm_bwd_da": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp],
    "dgg_softk_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _vp, _vp, _vp],
    "dgg_softk_bwd_rows": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp],
    "dgg_part_ws_bytes": [_i64, _i32, _i64],
    "dgg_part_build": [_vp, _vp, _i64, _i32, _i64, _vp, _vp],
    "dgg_edge_bwd_part": [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _vp, _i64, _vp, _vp, _vp],
    "dgg_softk_edge_bwd_part": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _i32, _i32, _vp, _i64, _vp, _vp,
                                _vp, _vp, _vp],
    "dgg_ell_conv_bwd_part": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "dgg_partp_ws_bytes": [_i64, _i32, _i64],
    "dgg_partp_has_map": [_i64],
    "dgg_partp_describe": [_i64, _i32, _i64, _vp],
    "dgg_partp_build_phase": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _i32, _vp],
    "dgg_softk_edge_bwd_partp_phase": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _i32, _i32, _vp, _i64,
                                       _vp, _vp, _vp, _i32, _i32, _vp],
    "dgg_partp_build": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp],
    "dgg_partp_build_norm": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp],
    "dgg_ell_conv_bwd_partp": [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_ell_conv_bwd_partp_ext": [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dgg_softk_edge_bwd_partp": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _i32, _i32, _vp, _i64,
                                 _vp, _vp, _vp, _i32, _vp],
    "dgg_pack_bf16": [_vp, _i64, _i64, _i32, _vp, _i64, _vp],
    "dgg_pack_bf16_both": [_vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp],
    "dgg_gemm_nt_bf16": [_vp, _vp, _i64, _i64, _i64, _f32, _vp, _vp],
    "dgg_gcnii_gemm_bf16": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _f32, _vp, _vp],
    "dgg_gcnii_gemm_bf16_split": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _f32, _vp, _vp],
    "dgg_gemm_nt_bf16_rows2": [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _f32, _vp, _vp],
    "dgg_gcnii_dsupport_bf16": [_vp, _vp, _i64, _i64, _vp, _f32, _f32, _vp, _vp, _vp],
    "dgg_ell_spmm_fwd_bf16": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp],
    "dgg_gcnii_stack_epilogue": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _u32, _u32, _vp, _vp, _vp],
    "dgg_gcnii_gemm_bf16_split_act": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f32, _f32, _i32, _f32, _u32, _u32, _vp, _vp, _vp],
    "dgg_dropout_hash": [_vp, _i64, _f32, _u32, _u32, _i32, _vp, _vp, _vp],
    "dgg_ell_spmm_fwd_b16": [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp],
    "dgg_ell_sddmm_b16": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp],
    "dgg_ell_sddmm_b16_ws_floats": [_i64, _i32, _i32],
    "dgg_ell_sddmm_slices_sum": [_vp, _i64, _i32, _i32, _vp, _i32, _vp],
    "dgg_ell_sddmm_b16_sliced": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "dgg_ell_spmm_t_part_b16": [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp],
    "dgg_gcnii_dsupport_bf16_b": [_vp, _vp, _i64, _i64, _vp, _f32, _f32, _vp, _vp, _vp, _i32, _vp],
    "dgg_gcnii_gout_pack": [_vp, _vp, _f32, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp],
    "dgg_ell_spmm_act_fwd": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp],
    "dgg_act_bwd": [_vp, _vp, _i64, _i32, _vp, _vp],
    "dgg_norm_bwd_da_part": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _i64, _vp, _vp, _vp],
    "dgg_gcnii_epilogue_fwd": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp],
    "dgg_gcnii_epilogue_bwd": [_vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp],
    "dgg_ell_spmm_t_part": [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp],
    "dgg_norm_da_cols_part": [_vp, _i64, _i32, _i64, _vp, _vp, _vp],
    "dgg_ell_sddmm_norm_part": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp],
    "dgg_edge_bwd": [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _vp, _vp],
    "dgg_edge_bwd_wide_rows": [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _vp, _vp, _vp],
    "dgg_edge_bwd_wide_rows_ws_floats": [_i64, _i32, _i32],
    "dgg_edge_bwd_wide_rows_sliced": [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _vp, _vp, _vp, _vp],
}

_lib = None


class DggHipError(RuntimeError):
    pass


def lib():
    """Loads libdgg_hip.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise DggHipError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
        L = C.CDLL(SO_PATH)
        L.dgg_last_error.restype = C.c_char_p
        L.dgg_abi_version.restype = C.c_int
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(L, name)      # AttributeError if the library does not export a declared symbol
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name in ("dgg_allpairs_workspace_bytes", "dgg_allpairs_anywide_ws_bytes", "dgg_allpairs_rowmin_ws_bytes", "dgg_allpairs_sweep_ctl_offset_bytes", "dgg_allpairs_rsym_ctl_offset_bytes", "dgg_gemm_tn_ws_floats", "dgg_gemm_tn_multi_ws_floats", "dgg_linear_bwd_ws_floats", "dgg_part_ws_bytes", "dgg_partp_ws_bytes", "dgg_knet_x_bwd_ws_bytes",
                     "dgg_degree_stats_ws_bytes", "dgg_ell_sddmm_b16_ws_floats", "dgg_edge_bwd_wide_rows_ws_floats"):
            getattr(L, name).restype = C.c_size_t
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        raise DggHipError(f"{what} failed (code {code}): {lib().dgg_last_error().decode()}")
