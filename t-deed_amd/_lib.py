"""ctypes binding of csrc/libtdeed_hip.so (C ABI declared in include/tdeed_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, an exception is
raised.  ``import torch`` happens first so that the library binds to the same libamdhip64 that
PyTorch-ROCm already loaded (same SONAME) and streams / device pointers are interchangeable.
"""
import ctypes
import os
from ctypes import c_int, c_long, c_float, c_void_p, c_char_p, c_uint64, POINTER

import torch  # noqa: F401  (must precede CDLL: see module docstring)

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libtdeed_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


P = c_void_p
_SIGS = {
    "tdeed_abi_version": ([], c_int),
    "tdeed_device_info": ([c_int, c_char_p, POINTER(c_int), POINTER(c_int)], c_int),
    "tdeed_stem_fwd": ([P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, c_int, P], c_int),
    "tdeed_augment_scratch_floats": ([c_int], c_long),
    "tdeed_augment_clips": ([P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P], c_int),
    "tdeed_mix_frames": ([P, P, P, c_int, c_long, P, P], c_int),
    "tdeed_avgpool_posenc_bwd": ([P, c_int, c_int, c_int, c_int, P, P, c_int, P], c_int),
    "tdeed_stem_mfma_parts": ([c_int, c_int], c_int),
    "tdeed_stem_mfma_fwd": ([P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P], c_int),
    "tdeed_stem_wgrad": ([P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P], c_int),
    "tdeed_stem_wgrad_bn_fits": ([c_int, c_int, c_int, c_int], c_int),
    "tdeed_stem_wgrad_bn": ([P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P], c_int),
    "tdeed_s1_front_parts": ([c_int, c_int, c_int], c_int),
    "tdeed_s1_front_fwd": ([P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P, P, P, P, P,
                            P, P, P, P, P, P, P], c_int),
    "tdeed_gemm_fwd": ([P, c_long, P, c_long, c_int, P, c_int, c_int, c_int, c_int, P, c_long, P, P, P, c_long,
                        c_int, P, c_long, c_int, c_int, c_int, c_int, c_int, P, P, c_long, c_int, c_int, c_int, P], c_int),
    "tdeed_gemm_ws_fits": ([c_int, c_int, c_int], c_int),
    "tdeed_gemm_ws_fwd": ([P, c_long, P, c_long, c_int, P, c_int, c_int, c_int, c_int, P, P, P, P, c_long,
                           c_int, P, c_long, c_int, c_int, c_int, c_int, c_int, P, c_long, c_int, c_int, P], c_int),
    "tdeed_gconv3x3_parts": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "tdeed_gconv3x3_mfma_fits": ([c_int, c_int, c_int, c_int], c_int),
    "tdeed_gconv3x3_fwd": ([P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, P], c_int),
    "tdeed_gemm_splitk_splits": ([c_int], c_int),
    "tdeed_gemm_splitk_fwd": ([P, c_long, c_int, c_int, c_int, P, c_long, P, P, P, c_long, c_int, P, c_long, P, P], c_int),
    "tdeed_se_gate_mfma_fits": ([c_int, c_int], c_int),
    "tdeed_se_gate_mfma_fwd": ([P, c_int, c_float, c_int, c_int, c_int, P, P, P, P, P, P], c_int),
    "tdeed_gemm_rs_fits": ([c_int, c_int, c_int], c_int),
    "tdeed_gemm_rs_grid": ([c_int], c_int),
    "tdeed_gemm_rs_stats_fwd": ([P, c_long, P, c_long, c_int, c_int, c_int, c_int, P, P, c_long, P, P], c_int),
    "tdeed_gemm_rs_fwd": ([P, c_long, P, c_long, c_int, P, c_int, c_int, c_int, c_int, P, P, P, P, c_long, c_int, P, c_long, P,
                           c_long, c_int, P], c_int),
    "tdeed_c1_gconv_set_debug": ([P], c_int),
    "tdeed_c1_gconv_fits": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "tdeed_c1_gconv_slab_tiles": ([c_int, c_int, c_int, c_int], c_int),
    "tdeed_c1_gconv_fwd": ([P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P,
                            P], c_int),
    "tdeed_bneck_fits": ([c_int, c_int, c_int, c_int], c_int),
    "tdeed_bneck_set_debug": ([P], c_int),
    "tdeed_bneck_fwd": ([P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P, P, P, P, P, c_int,
                         c_int, P], c_int),
    "tdeed_bneck_gs_fwd": ([P, P, c_int, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P,
                            P, P, P, c_int, P, P, P, P, P, c_int, c_int, P, P, c_int, P, P], c_int),
    "tdeed_bneck_qtail_fits": ([c_int, c_int, c_int, c_int], c_int),
    "tdeed_gsf_gate_sums_fwd": ([P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P], c_int),
    "tdeed_bn_slabs": ([c_long], c_int),
    "tdeed_bn_train_stats": ([P, c_long, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P, c_int, P], c_int),
    "tdeed_bn_apply": ([P, c_long, c_int, P, P, P, c_int, P, c_int, P], c_int),
    "tdeed_bn_apply_slice": ([P, c_long, c_int, P, P, P, P, P, c_int, P, P, c_int, c_int, c_int, P], c_int),
    "tdeed_bn_apply2": ([P, c_long, c_int, P, P, P, P, P, c_int, P, c_int, P], c_int),
    "tdeed_bn_train_bwd": ([P, P, P, c_int, c_long, c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, P], c_int),
    "tdeed_fold_rows": ([P, c_long, c_int, c_int, c_int, P, c_long, P], c_int),
    "tdeed_bn_finalize": ([P, P, c_long, c_int, c_long, c_int, P, P, c_float, c_float, P, P, P, P, P, P, P], c_int),
    "tdeed_pool_rows": ([P, P, c_int, c_int, c_int, P, P, c_int, P, c_int, P], c_int),
    "tdeed_se_train_fwd": ([P, c_int, c_int, c_int, P, P, P, P, P, P, P], c_int),
    "tdeed_se_train_bwd": ([P, P, P, c_int, c_int, c_int, P, P, P, P, P, P], c_int),
    "tdeed_scale_rows": ([P, P, P, c_float, c_int, c_int, c_int, P, P, P, c_int, P], c_int),
    "tdeed_gconv3x3_dgrad_stats": ([P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P], c_int),
    "tdeed_gconv3x3_bwd_stats_bands": ([c_int], c_int),
    "tdeed_gconv3x3_bwd_stats_fits": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "tdeed_gconv3x3_bwd_stats": ([P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P], c_int),
    "tdeed_bn_bwd_masked_from_parts": ([P, P, c_long, c_int, P, P, P, P, P, P, P, c_long, c_int, P, P, c_int, P], c_int),
    "tdeed_narrow_conv1_bwd_fits": ([c_int, c_int], c_int),
    "tdeed_narrow_conv1_bwd_grid": ([c_long, c_int, c_int], c_int),
    "tdeed_narrow_conv1_bwd": ([P, P, c_long, c_int, c_int, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, P, c_int, P, P, P, P,
                                P, P, P], c_int),
    "tdeed_bn_sums_from_parts": ([P, P, c_long, c_int, c_int, P, P, P], c_int),
    "tdeed_se_bn_bwd_sums": ([P, P, c_int, c_int, c_int, P, P, P, P, c_int, P], c_int),
    "tdeed_se_bn_bwd_finalize": ([P, P, P, c_int, c_int, c_int, P, P, P], c_int),
    "tdeed_se_bn_bwd_apply": ([P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int, P], c_int),
    "tdeed_gemm_dgrad": ([P, c_long, c_int, c_int, c_int, P, c_long, P, c_long, c_int, c_int, P, c_long, P, c_long, c_int, P,
                          c_long, P, c_long, P, P, c_long, P, P, c_int, P], c_int),
    "tdeed_gemm_dgrad_rs": ([P, c_long, c_int, c_int, c_int, P, P, c_long, P, c_long, P, c_long, c_int, P, c_long, P, c_long, P, P,
                             P], c_int),
    "tdeed_gsf_add_cols_sink_parts": ([c_long, c_int, c_int], c_int),
    "tdeed_gsf_add_cols_sink": ([P, P, c_long, c_int, c_int, P, P, c_long, P, c_long, P, P, c_long, P, P, c_int, P], c_int),
    "tdeed_bn_bwd_from_parts": ([P, P, c_long, c_int, P, P, P, P, c_int, P, c_int, c_int, c_int, P, P, P, c_int, P], c_int),
    "tdeed_gconv_wgrad_slabs": ([c_long], c_int),
    "tdeed_gconv3x3_bwd": ([P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, P], c_int),
    "tdeed_stride2_rows": ([P, P, c_int, c_int, c_int, c_int, c_int, c_int, P], c_int),
    "tdeed_loss2": ([P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, c_float, P, P, P], c_int),
    "tdeed_reduce_strided": ([P, c_int, c_long, c_long, P, P], c_int),
    "tdeed_gsf_slice": ([P, c_long, c_int, c_int, c_int, P, c_int, P], c_int),
    "tdeed_gsf_bwd_scratch_floats": ([c_int, c_int, c_int, c_int], c_long),
    "tdeed_gsf_bwd_part_layout": ([c_int, c_int, c_int, c_int, P], c_int),
    "tdeed_sgp_gemm_ksteps": ([c_int], c_int),
    "tdeed_sgp_gemm_row_tiles": ([c_int, c_int], c_int),
    "tdeed_sgp_gemm_col_tiles": ([c_int, c_int], c_int),
    "tdeed_sgp_gemm_form": ([c_int, c_int, c_int, c_int, c_int], c_int),
    "tdeed_sgp_gemm_gn_gelu": ([P, c_int, c_int, c_int, P, c_int, P, P, c_int, c_float, P, P, c_int, P, c_int, c_int, P], c_int),
    "tdeed_sgp_gemm_residual": ([P, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, P], c_int),
    "tdeed_sgp_gemm_gelu_chsum": ([P, c_int, c_int, c_int, P, P, c_int, P, P, P, c_int, c_int, P], c_int),
    "tdeed_gsf_bwd": ([P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P,
                       c_int, P], c_int),
    "tdeed_gsf_add_cols": ([P, P, c_long, c_int, c_int, P, c_int, P], c_int),
    "tdeed_gsf_bwd_bn_parts": ([c_int, c_int, c_int, c_int, c_int, c_int], c_int),
    "tdeed_gsf_bwd_stats": ([P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P,
                             P, P, c_int, P], c_int),
    "tdeed_gsf_add_cols_sink_bn": ([P, P, c_long, c_int, c_int, P, P, c_long, P, c_long, P, P, c_long, P, P, P, P, P, P, P, c_int,
                                    P], c_int),
    "tdeed_multi_fold_cw": ([c_int, c_long], c_int),
    "tdeed_multi_fold": ([P, c_int, c_long, P, c_float, c_int, P], c_int),
    "tdeed_multi_copy": ([P, c_int, c_long, P, c_float, c_int, P], c_int),
    "tdeed_reduce_partials": ([P, c_int, c_long, P, c_int, P], c_int),
    "tdeed_eltwise": ([P, P, P, c_long, c_int, c_int, P], c_int),
    "tdeed_transpose": ([P, c_int, c_int, P, c_int, P], c_int),
    "tdeed_wgrad_slices": ([c_int, c_int, c_int], c_int),
    "tdeed_wgrad": ([P, c_long, P, c_long, P, c_long, c_int, c_int, c_int, c_int, P, P, P, P, c_int, c_int, P], c_int),
    "tdeed_layernorm_bwd_blocks": ([c_int], c_int),
    "tdeed_layernorm_bwd": ([P, c_long, P, c_long, c_int, c_int, P, c_float, P, c_int, P, P, P, c_int, P], c_int),
    "tdeed_groupnorm_bwd": ([P, P, c_int, c_int, c_int, c_int, P, c_float, P, c_int, P, P, P, c_int, P], c_int),
    "tdeed_sgp_branch_bwd": ([P, c_long, P, P, P, c_long, c_int, c_int, c_int, c_int, c_int, P, P, P, c_long, P, P, P, P,
                              c_int, P], c_int),
    "tdeed_upsample_bwd": ([P, c_long, c_int, c_int, c_int, c_int, P, c_int, P], c_int),
    "tdeed_maxpool_bwd": ([P, P, c_int, c_int, c_int, c_int, P, c_int, P], c_int),
    "tdeed_se_gate_fwd": ([P, c_int, c_float, c_int, c_int, c_int, P, P, P, P, P, P], c_int),
    "tdeed_se_gate_bf16_fwd": ([P, c_int, c_float, c_int, c_int, c_int, P, P, P, P, P, P], c_int),
    "tdeed_gsf_gate_fwd": ([P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_int, P], c_int),
    "tdeed_gsf_weight_fwd": ([P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P], c_int),
    "tdeed_gsf_apply_fwd": ([P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P], c_int),
    "tdeed_gsf_apply_fused_fwd": ([P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P], c_int),
    "tdeed_gsf_blend_src_fwd": ([P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P], c_int),
    "tdeed_avgpool_posenc_fwd": ([P, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, P], c_int),
    "tdeed_layernorm_fwd": ([P, c_long, c_int, c_int, P, P, c_float, P, c_long, c_int, P], c_int),
    "tdeed_sgp_branch_fwd": ([P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P], c_int),
    "tdeed_mixer_branch_fwd": ([P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P], c_int),
    "tdeed_sgp_front_set_debug": ([P], c_int),
    "tdeed_sgp_front_fwd": ([P, c_int, c_int, c_int, c_int, c_int, P, P, c_float, P, P, P, P, P, c_int, P, c_int, P], c_int),
    "tdeed_mixer_front_fwd": ([P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_float, P, P, P, P, P, P, c_int, P,
                               c_int, c_int, c_int, P], c_int),
    "tdeed_gemm_splitk_partials": ([P, c_long, c_int, c_int, c_int, P, c_long, P, P], c_int),
    "tdeed_groupnorm_fwd": ([P, c_int, c_int, c_int, c_int, P, P, c_float, P, c_int, P], c_int),
    "tdeed_maxpool_fwd": ([P, c_int, c_int, c_int, c_int, P, c_int, P], c_int),
    "tdeed_maxpool_rowstat_fwd": ([P, c_int, c_int, c_int, c_int, P, P, c_float, c_int, P], c_int),
    "tdeed_heads_fwd": ([P, c_int, c_int, P, P, c_int, P, c_int, P], c_int),
    "tdeed_loss_fwd": ([P, c_int, c_int, c_int, P, P, P, c_int, P, P, P], c_int),
    "tdeed_loss_bwd": ([P, c_int, c_int, c_int, P, P, P, c_int, P, c_float, P, P], c_int),
    "tdeed_heads_bwd_workspace": ([c_int, c_int, c_int], c_long),
    "tdeed_heads_bwd": ([P, P, c_int, c_int, P, c_int, P, P, P, P, c_int, P], c_int),
    "tdeed_adamw_step": ([P, P, P, P, c_long, c_float, c_float, c_float, c_float, c_float, c_int, c_float, P], c_int),
    "tdeed_process_prediction": ([P, c_int, c_int, c_int, c_int, c_int, P, P, P], c_int),
    "tdeed_gather_cast": ([P, P, c_long, P, c_int, P], c_int),
    "tdeed_cast_f32_to_bf16": ([P, P, c_long, P], c_int),
    "tdeed_multi_cast_transpose": ([P, c_int, c_long, c_int, P], c_int),
    "tdeed_fill_u8_hash": ([P, c_long, c_uint64, P], c_int),
    "tdeed_comm_unique_id": ([P], c_int),
    "tdeed_comm_init": ([POINTER(c_void_p), P, c_int, c_int], c_int),
    "tdeed_comm_info": ([P, POINTER(c_int), POINTER(c_int)], c_int),
    "tdeed_comm_all_reduce": ([P, P, c_long, c_int, P], c_int),
    "tdeed_comm_all_reduce_rs_ag": ([P, P, c_long, c_int, P], c_int),
    "tdeed_comm_join": ([P, P], c_int),
    "tdeed_comm_destroy": ([P], c_int),
    "tdeed_graph_begin": ([P], c_int),
    "tdeed_graph_end": ([P, POINTER(c_void_p)], c_int),
    "tdeed_graph_launch": ([P, P], c_int),
    "tdeed_graph_destroy": ([P], c_int),
}
EXPORTS = tuple(_SIGS) + ("tdeed_last_error",)

_lib = None


def load():
    """Load (once) and return the ctypes library; raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    flavour = os.environ.get("TDEED_LIB_FLAVOUR", "release")
    if flavour == "debug":
        # the asserting build (csrc/common.h TD_DEV_ASSERT): `python t-deed_amd/build.py --debug`
        path = LIB_PATH.replace("libtdeed_hip.so", "libtdeed_hip_dbg.so")
    elif flavour != "release":
        # an A/B build of the same ABI with some sources taken from another revision (tools/build_ab.py): lets one gpurun
        # call time two forms of a kernel on the SAME box (boxes differ by 2-3 %)
        path = LIB_PATH.replace("libtdeed_hip.so", f"libtdeed_hip_{flavour}.so")
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950; the debug flavour with `python t-deed_amd/build.py --debug`).  "
            "tdeed_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (args, res) in _SIGS.items():
        if flavour not in ("release", "debug") and not hasattr(lib, name):
            continue                                 # an A/B flavour built from an older revision may lack a newer entry point
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    lib.tdeed_last_error.argtypes = []
    lib.tdeed_last_error.restype = c_char_p
    _lib = lib
    return lib


_PROFILE = None      # list of (entry name, start event, end event, args) while a `profile()` block is open
SCOPE = ""           # label the training engine sets per stage / block ("s3.b2.bwd"): recorded with every profiled call


class profile:
    """Device time per C-ABI entry point: inside the block every `call()` is bracketed by two HIP events on the current
    stream; `.summary()` (after the block) synchronises and returns {name: dict(ms, calls, args=[...])}.  Measurement
    plumbing for bench.py's per-family roofline of the training step (the product path never opens one)."""

    def __enter__(self):
        global _PROFILE
        self.rec = _PROFILE = []
        return self

    def __exit__(self, *exc):
        global _PROFILE
        _PROFILE = None
        return False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b, args, *_ in self.rec:
            d = out.setdefault(name, dict(ms=0.0, calls=0, args=[]))
            d["ms"] += a.elapsed_time(b)
            d["calls"] += 1
            d["args"].append(args)
        return out

    def by_scope(self):
        """{scope: {entry: ms}} (scope = _lib.SCOPE when the call was issued)"""
        torch.cuda.synchronize()
        out = {}
        for name, a, b, args, scope in self.rec:
            d = out.setdefault(scope, {})
            d[name] = d.get(name, 0.0) + a.elapsed_time(b)
        return out


def call(name, *args):
    lib = load()
    if _PROFILE is not None and name not in ("tdeed_graph_begin", "tdeed_graph_end", "tdeed_graph_launch"):
        st = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        rc = getattr(lib, name)(*args)
        b.record(st)
        _PROFILE.append((name, a, b, args, SCOPE))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HipCallError(f"{name} -> {rc}: {lib.tdeed_last_error().decode(errors='replace')}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def dtype_code(torch_dtype):
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported activation dtype {torch_dtype}")
