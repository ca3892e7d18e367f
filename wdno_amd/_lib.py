"""ctypes binding of libwdno_hip.so (the C ABI declared in include/wdno_hip.h).

The product path has NO fallback: if the shared library is missing or a call returns an error, a RuntimeError is
raised. Nothing in this package computes on the CPU.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first: libwdno_hip.so has to bind to the HIP runtime instance torch already loaded

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libwdno_hip.so')

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
Z = C.c_size_t


class DwtDesc(C.Structure):
    _fields_ = [('nd', I), ('mode', I), ('L', I), ('n_img', I), ('in_dims', I * 3), ('out_dims', I * 3),
                ('cs_img', L), ('cs_band', L), ('cs0', L), ('cs1', L)]


class ConvGeom(C.Structure):
    _fields_ = [(k, I) for k in ('N', 'D', 'H', 'W', 'C', 'OD', 'OH', 'OW', 'K', 'kd', 'kh', 'kw', 'sd', 'sh', 'sw',
                                 'pd', 'ph', 'pw', 'YD', 'YH', 'YW', 'osd', 'osh', 'osw', 'ood', 'ooh', 'oow')]


class AttnDesc(C.Structure):
    _fields_ = [('n_uo', I), ('n_ui', I), ('n_tok', I), ('heads', I), ('so', L), ('si', L), ('st', L)]


class CondDesc(C.Structure):
    _fields_ = [(k, I) for k in ('tree', 'B', 'F', 'C', 'H', 'W', 'cT', 'cH', 'cW', 'cond_pad', 'cond_a', 'cond_b',
                                 'cond_c', 'cond_low', 'u_rows', 'uT_rows')]


class WgradReduceItem(C.Structure):          # include/wdno_hip.h: wdno_wgrad_reduce_item
    _fields_ = [('ws', C.c_void_p), ('dw', C.c_void_p), ('n', I), ('splits', I), ('K8', I), ('C8', I), ('kw', I), ('ntap', I), ('Kn', I), ('Cn', I),
                ('tiled', I), ('reserved', I)]


class RowsSumItem(C.Structure):              # include/wdno_hip.h: wdno_rows_sum_item
    _fields_ = [('part', C.c_void_p), ('out', C.c_void_p), ('rows', I), ('stride', I), ('col0', I), ('ncols', I), ('is_double', I), ('reserved', I)]


PD, PG, PA, PC = C.POINTER(DwtDesc), C.POINTER(ConvGeom), C.POINTER(AttnDesc), C.POINTER(CondDesc)
PF = C.POINTER(C.c_float)

# name -> (restype, argtypes). Every symbol declared in include/wdno_hip.h appears here (tests check both directions).
PROTOTYPES = {
    'wdno_strerror': (C.c_char_p, [I]),
    'wdno_version': (I, []),
    'wdno_last_hip_error': (C.c_char_p, []),
    'wdno_set_debug': (I, [I]),
    'wdno_dwt_ws_bytes': (Z, [PD]),
    'wdno_dwt_fwd': (I, [P, P, PD, PF, P, Z, P]),
    'wdno_dwt_fwd_packed': (I, [P, P, PD, PF, I, L, I, P, P]),
    'wdno_dwt_inv': (I, [P, P, PD, PF, P, Z, P]),
    'wdno_dwt_fwd_adjoint': (I, [P, P, PD, PF, P, Z, P]),
    'wdno_dwt_inv_adjoint': (I, [P, P, PD, PF, P, Z, P]),
    'wdno_upsample_coef': (I, [P, P, L, I, I, I, I, I, I, I, P]),
    'wdno_pack_smoke_state': (I, [P, L, P, L, P, L, P, P, P, L, I, I, I, I, I, P]),
    'wdno_pack_smoke_fields': (I, [P, L, P, L, P, L, PF, PF, I, P, P, L, I, I, I, I, I, I, I, I, P]),
    'wdno_nc_to_cl': (I, [P, P, L, I, L, I, P]),
    'wdno_cl_to_nc': (I, [P, P, L, I, L, I, P]),
    'wdno_concat2_cl': (I, [P, I, P, I, P, L, P]),
    'wdno_concat2_cl_planes': (I, [P, I, P, I, P, P, P, P, P, L, P]),
    'wdno_concat2_cl_amax': (I, [P, I, P, I, P, P, L, P]),
    'wdno_split2_cl': (I, [P, P, I, P, I, L, P]),
    'wdno_upsample2x_cl_fwd': (I, [P, P, L, I, I, I, P]),
    'wdno_upsample2x_cl_bwd': (I, [P, P, L, I, I, I, P]),
    'wdno_conv_fwd': (I, [P, P, P, P, P, PG, P]),
    'wdno_amax': (I, [P, L, P, P]),
    'wdno_amax_record': (I, [P, L, P, P]),
    'wdno_split_f16': (I, [P, P, P, P, P, L, I, I, P]),
    'wdno_split_colsum_ws_bytes': (Z, [L, I]),
    'wdno_split_f16_colsum': (I, [P, P, P, P, P, P, P, Z, L, I, I, P]),
    'wdno_pack_split_weight': (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    'wdno_amax_multi': (I, [P, I, I, P]),
    'wdno_pack_split_weight_multi': (I, [P, I, I, P]),
    'wdno_conv_fwd_f16x3': (I, [P, P, P, P, P, P, P, P, P, PG, P]),
    'wdno_conv_fwd_f16x3_amax': (I, [P, P, P, P, P, P, P, P, P, P, PG, P]),
    'wdno_conv_fwd_f16x3_ws': (I, [P, P, P, P, P, P, P, P, P, P, PG, P, Z, P]),
    'wdno_conv_fwd_split_ws_bytes': (Z, [PG]),
    'wdno_conv_wgrad_f16x3_ws_bytes': (Z, [PG]),
    'wdno_conv_pixel_table': (I, [P, PG, P]),
    'wdno_conv_wgrad_f16x3': (I, [P, P, P, P, P, P, P, P, P, Z, PG, P]),
    'wdno_conv_wgrad_f16x3_param': (I, [P, P, P, P, P, P, P, P, I, I, P, Z, PG, P]),
    'wdno_conv_wgrad_partials': (I, [P, P, P, P, P, P, P, P, I, I, P, Z, PG, P, P]),
    'wdno_wgrad_reduce_multi': (I, [P, I, P]),
    'wdno_rows_sum_multi': (I, [P, I, P]),
    'wdno_cast_bf16': (I, [P, P, L, I, I, P]),
    'wdno_cast_bf16_colsum': (I, [P, P, P, P, Z, L, I, I, P]),
    'wdno_conv_fwd_bf16': (I, [P, P, P, P, P, P, PG, P]),
    'wdno_conv_fwd_f16x3_zbox': (I, [P, P, P, P, P, P, P, P, P, P, PG, P, P]),
    'wdno_conv_fwd_bf16_ex': (I, [P, P, P, P, P, I, P, PG, P, P]),
    'wdno_conv_wgrad_bf16_param': (I, [P, P, P, P, I, I, P, Z, PG, P]),
    'wdno_conv_wgrad_ws_bytes': (Z, [PG]),
    'wdno_conv_wgrad': (I, [P, P, P, P, Z, PG, P]),
    'wdno_colsum_ws_bytes': (Z, [L, I]),
    'wdno_colsum': (I, [P, P, L, I, P, Z, P]),
    'wdno_groupnorm_ws_bytes': (Z, [L, L, I, I]),
    'wdno_groupnorm_stats_floats': (Z, [L, I, I]),
    'wdno_groupnorm_act_fwd': (I, [P, P, P, P, P, P, L, L, I, I, F, I, P, Z, P]),
    'wdno_groupnorm_act_fwd_amax': (I, [P, P, P, P, P, P, P, L, L, I, I, F, I, P, Z, P]),
    'wdno_groupnorm_act_bwd': (I, [P, P, P, P, P, P, P, P, P, L, L, I, I, I, P, Z, P]),
    'wdno_groupnorm_act_bwd_amax': (I, [P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, P, Z, P]),
    'wdno_groupnorm_bwd_planes_ws_bytes': (Z, [L, L, I, I]),
    'wdno_groupnorm_bwd_planes_tail': (I, [L, L, I, I, C.POINTER(Z), C.POINTER(I)]),
    'wdno_groupnorm_fwd_planes_ws_bytes': (Z, [L, L, I, I]),
    'wdno_groupnorm_act_fwd_planes': (I, [P, P, P, P, P, P, P, P, P, L, L, I, I, F, I, P, Z, P]),
    'wdno_groupnorm_act_add_fwd_planes': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, F, I, P, Z, P]),
    'wdno_layernorm_fwd_planes': (I, [P, P, P, P, P, L, I, F, P]),
    'wdno_groupnorm_act_bwd_planes': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, P, Z, P]),
    'wdno_groupnorm_act_fwd_amax_t': (I, [P, I, P, P, P, P, P, P, L, L, I, I, F, I, P, Z, P]),
    'wdno_groupnorm_act_fwd_planes_t': (I, [P, I, P, P, P, P, P, P, P, P, L, L, I, I, F, I, P, Z, P]),
    'wdno_groupnorm_act_add_fwd_planes_t': (I, [P, I, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, F, I, P, Z, P]),
    'wdno_groupnorm_act_bwd_planes_t': (I, [P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, P, Z, P]),
    'wdno_layernorm_fwd': (I, [P, P, P, L, I, F, P]),
    'wdno_layernorm_fwd_amax': (I, [P, P, P, P, L, I, F, P]),
    'wdno_layernorm_bwd_ws_bytes': (Z, [L, I]),
    'wdno_layernorm_bwd': (I, [P, P, P, P, P, L, I, F, P, Z, P]),
    'wdno_layernorm_bwd_add': (I, [P, P, P, P, P, P, L, I, F, P, Z, P]),
    'wdno_layernorm_bwd_add_amax': (I, [P, P, P, P, P, P, P, L, I, F, P, Z, P]),
    'wdno_attn_fwd': (I, [P, P, P, P, P, PA, F, P]),
    'wdno_attn_bwd_ws_bytes': (Z, [PA]),
    'wdno_attn_bwd': (I, [P, P, P, P, P, P, P, P, PA, F, P, Z, P]),
    'wdno_attn_fwd_amax': (I, [P, P, P, P, P, P, PA, F, P]),
    'wdno_attn_bwd_amax': (I, [P, P, P, P, P, P, P, P, P, PA, F, P, Z, P]),
    'wdno_attn_fwd_planes': (I, [P, P, P, P, P, P, P, P, P, P, PA, F, P]),
    'wdno_linattn_fwd_planes': (I, [P, P, P, P, P, P, P, L, I, I, F, P]),
    'wdno_attn_bwd_planes': (I, [P, P, P, P, P, P, P, P, P, P, P, P, PA, F, P, Z, P]),
    'wdno_tattn_fused_takes': (I, [I, I, I]),
    'wdno_tattn_fused_fwd': (I, [P, P, F, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, L, I, I, F, P]),
    'wdno_lattn_fused_takes': (I, [I, I, I]),
    'wdno_lattn_fused_ws_bytes': (Z, [L, I]),
    'wdno_lattn_fused_fwd': (I, [P, P, F, P, P, P, P, P, P, P, P, P, P, P, P, Z, L, I, I, I, F, P]),
    'wdno_lattn_fused_bwd_grads': (I, []),
    'wdno_lattn_fused_bwd_ws_bytes': (Z, [L, I]),
    'wdno_lattn_fused_bwd': (I, [P, P, P, F, P, P, P, P, P, P, P, P, P, P, P, P, P, Z, L, I, I, I, F, P]),
    'wdno_tattn_fused_bwd_ws_bytes': (Z, []),
    'wdno_tattn_fused_bwd_grads': (I, []),
    'wdno_tattn_fused_bwd': (I, [P, P, P, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, Z, L, I, L, I, I, F, P]),
    'wdno_linattn_ws_bytes': (Z, [L, I]),
    'wdno_linattn_fwd': (I, [P, P, P, P, L, I, I, F, P]),
    'wdno_linattn_bwd': (I, [P, P, P, P, P, P, Z, L, I, I, F, P]),
    'wdno_linattn_fwd_amax': (I, [P, P, P, P, P, L, I, I, F, P]),
    'wdno_linattn_bwd_amax': (I, [P, P, P, P, P, P, P, Z, L, I, I, F, P]),
    'wdno_linattn_bwd_planes': (I, [P, P, P, P, P, P, P, P, P, P, P, Z, L, I, I, F, P]),
    'wdno_act_fwd': (I, [P, P, L, I, P]),
    'wdno_act_bwd': (I, [P, P, P, L, I, P]),
    'wdno_linear_rows_fwd': (I, [P, I, P, I, P, P, I, I, I, I, P]),
    'wdno_linear_rows_wgrad': (I, [P, I, P, I, P, P, I, I, I, P]),
    'wdno_linear_multi_fwd': (I, [P, I, I, P, P, I, I, P]),
    'wdno_linear_multi_wgrad': (I, [P, I, I, P, P, P, P, I, I, P]),
    'wdno_linear_multi_dgrad_ws_bytes': (Z, [I, I, I]),
    'wdno_linear_multi_dgrad': (I, [P, I, I, P, P, I, I, P, Z, P]),
    'wdno_add': (I, [P, P, P, L, P]),
    'wdno_add_amax': (I, [P, P, P, P, L, P]),
    'wdno_sinusoidal_emb': (I, [P, P, P, I, I, P]),
    'wdno_q_sample_cond': (I, [P, P, P, P, P, P, P, PC, P]),
    'wdno_apply_cond': (I, [P, P, PC, P]),
    'wdno_weighted_mse_ws_bytes': (Z, [L]),
    'wdno_weighted_mse': (I, [P, P, P, P, F, P, P, L, L, I, L, P, Z, P]),
    'wdno_weighted_mse_bwd': (I, [P, P, P, P, F, P, P, L, L, I, L, P]),
    'wdno_p_sample_update': (I, [P, P, P, P, P, P, P, P, P, P, P, L, L, I, P]),
    'wdno_ddim_update': (I, [P, P, P, P, P, P, F, F, F, P, P, L, L, P]),
    'wdno_ddim_update_dev': (I, [P, P, P, P, P, P, P, P, P, L, L, P]),
    'wdno_sumsq_ws_bytes': (Z, [L]),
    'wdno_sumsq': (I, [P, L, P, P, Z, P]),
    'wdno_adam_clip_step': (I, [P, P, P, P, L, P, F, F, F, F, F, F, I, P]),
    'wdno_ema_update': (I, [P, P, L, F, P]),
    'wdno_gather_items': (I, [P, I, I, P]),
    'wdno_relpos_bias_fwd': (I, [P, P, P, I, I, P]),
    'wdno_relpos_bias_bwd': (I, [P, P, P, I, I, I, P]),
}

_lib = None


def load():
    """Load the shared library and attach prototypes. Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(there is no CPU / eager fallback for the wdno_amd hot path)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    if os.environ.get('WDNO_DEBUG'):          # kernel-selection switches for A/B measurements (see the `debug` notes in csrc/)
        lib.wdno_set_debug(int(os.environ['WDNO_DEBUG']))
    return lib


def check(code, what):
    if code != 0:
        lib = load()
        msg = lib.wdno_strerror(code).decode()
        hip = lib.wdno_last_hip_error().decode() if code == -2 else ''
        raise RuntimeError(f'{what} failed: {msg} ({code}) {hip}')
