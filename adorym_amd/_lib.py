"""ctypes binding of libadm.so (include/adm.h).  No fallback: if the HIP library is missing or
cannot be loaded this module raises -- the product path never runs on a CPU substitute."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ADM_LIB_PATH') or os.path.join(_HERE, 'libadm.so')   # ADM_LIB_PATH: experiment builds only

ADM_OK, ADM_ERR_INVALID, ADM_ERR_HIP, ADM_ERR_UNSUPPORTED, ADM_ERR_NOMEM = 0, -1, -2, -3, -4
DET_NONE, DET_FARFIELD, DET_FRESNEL = 0, 1, 2
LOSS_LSQ, LOSS_POISSON = 0, 1
FLAG_NONNEG, FLAG_ZERO_CH0, FLAG_ZERO_CH1 = 1, 2, 4
OPT_ADAM, OPT_GD, OPT_MOMENTUM = 0, 1, 2
P2P_HANDLE_BYTES, P2P_MAX_RANKS = 64, 16


class PlanDesc(C.Structure):
    _fields_ = [('obj_y', C.c_int32), ('obj_x', C.c_int32), ('obj_z', C.c_int32),
                ('probe_y', C.c_int32), ('probe_x', C.c_int32),
                ('pad_y0', C.c_int32), ('pad_y1', C.c_int32), ('pad_x0', C.c_int32), ('pad_x1', C.c_int32),
                ('binning', C.c_int32), ('n_modes', C.c_int32), ('sign_convention', C.c_int32),
                ('det_mode', C.c_int32), ('normalize_fft', C.c_int32), ('k1', C.c_float),
                ('h_re', C.POINTER(C.c_float)), ('h_im', C.POINTER(C.c_float)),
                ('hfree_re', C.POINTER(C.c_float)), ('hfree_im', C.POINTER(C.c_float)),
                ('loss_type', C.c_int32), ('poisson_multiplier', C.c_float), ('unknown_type', C.c_int32)]


class HoloDesc(C.Structure):
    _fields_ = [('ny', C.c_int32), ('nx', C.c_int32), ('n_dists', C.c_int32), ('lambda_nm', C.c_double),
                ('voxel_nm_y', C.c_double), ('voxel_nm_x', C.c_double), ('sign_convention', C.c_int32),
                ('unknown_type', C.c_int32), ('raw_intensity', C.c_int32), ('k1', C.c_float)]


class SmallParam(C.Structure):
    _fields_ = [('x', C.c_void_p), ('g', C.c_void_p), ('m', C.c_void_p), ('v', C.c_void_p), ('n', C.c_uint64), ('step_size', C.c_double),
                ('center_cols', C.c_int32), ('zero_grad', C.c_int32), ('pin', C.c_void_p), ('pin_n', C.c_uint64)]


class HoloAdam(C.Structure):
    _fields_ = [('m_obj', C.c_void_p), ('v_obj', C.c_void_p), ('step_obj', C.c_double),
                ('m_dists', C.c_void_p), ('v_dists', C.c_void_p), ('step_dists', C.c_double),
                ('m_affine', C.c_void_p), ('v_affine', C.c_void_p), ('step_affine', C.c_double),
                ('affine_pin', C.c_void_p), ('affine_pin_n', C.c_uint64),
                ('i_batch', C.c_int32), ('b1', C.c_double), ('b2', C.c_double), ('eps', C.c_double)]


SMALL_PARAMS_MAX = 6
_VP, _SZ, _I, _F, _D = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double

# name -> (restype, argtypes); must list every symbol include/adm.h declares
SIGNATURES = {
    'adm_version': (_I, []),
    'adm_last_error': (C.c_char_p, []),
    'adm_crash_line_set': (_I, [C.c_char_p, C.c_int]),
    'adm_device_count': (_I, []),
    'adm_mem_info': (_I, [_VP, C.POINTER(_SZ), C.POINTER(_SZ)]),
    'adm_ctx_create': (_I, [_I, _VP, C.POINTER(_VP)]),
    'adm_ctx_destroy': (_I, [_VP]),
    'adm_ctx_sync': (_I, [_VP]),
    'adm_ctx_stream': (_VP, [_VP]),
    'adm_ctx_device': (_I, [_VP]),
    'adm_ctx_fork': (_I, [_VP]),
    'adm_ctx_end_fork': (_I, [_VP]),
    'adm_ctx_join': (_I, [_VP]),
    'adm_malloc': (_I, [_VP, _SZ, C.POINTER(_VP)]),
    'adm_free': (_I, [_VP, _VP]),
    'adm_memset': (_I, [_VP, _VP, _I, _SZ]),
    'adm_h2d': (_I, [_VP, _VP, _VP, _SZ]),
    'adm_d2h': (_I, [_VP, _VP, _VP, _SZ]),
    'adm_d2d': (_I, [_VP, _VP, _VP, _SZ]),
    'adm_host_alloc': (_I, [_VP, _SZ, C.POINTER(_VP)]),
    'adm_host_free': (_I, [_VP, _VP]),
    'adm_d2h_async': (_I, [_VP, _VP, _VP, _SZ]),
    'adm_h2d_async': (_I, [_VP, _VP, _VP, _SZ]),
    'adm_event_sync': (_I, [_VP, _VP]),
    'adm_event_create': (_I, [_VP, C.POINTER(_VP)]),
    'adm_event_destroy': (_I, [_VP, _VP]),
    'adm_event_record': (_I, [_VP, _VP]),
    'adm_event_elapsed_ms': (_I, [_VP, _VP, _VP, C.POINTER(_F)]),
    'adm_comm_unique_id': (_I, [_VP]),
    'adm_comm_available': (_I, []),
    'adm_comm_init': (_I, [_VP, _I, _I, _VP]),
    'adm_comm_init_aux': (_I, [_VP, _VP]),
    'adm_comm_destroy': (_I, [_VP]),
    'adm_comm_rank': (_I, [_VP]),
    'adm_comm_size': (_I, [_VP]),
    'adm_reduce_scatter': (_I, [_VP, _VP, _VP, _SZ]),
    'adm_all_gather': (_I, [_VP, _VP, _VP, _SZ]),
    'adm_all_reduce': (_I, [_VP, _VP, _SZ, _I]),
    'adm_broadcast': (_I, [_VP, _VP, _SZ, _I]),
    'adm_reduce': (_I, [_VP, _VP, _SZ, _I]),
    'adm_comm_group_start': (_I, [_VP]),
    'adm_comm_group_end': (_I, [_VP]),
    'adm_p2p_create': (_I, [_VP, _I, _I, _SZ]),
    'adm_p2p_local': (_I, [_VP, _I, C.POINTER(_VP)]),
    'adm_p2p_export': (_I, [_VP, _VP, _VP]),
    'adm_p2p_open': (_I, [_VP, _VP, C.POINTER(_VP)]),
    'adm_p2p_close': (_I, [_VP, _VP]),
    'adm_p2p_connect': (_I, [_VP, C.POINTER(_VP), C.POINTER(_VP)]),
    'adm_p2p_bind_object': (_I, [_VP, C.POINTER(_VP), C.POINTER(_VP), _SZ]),
    'adm_p2p_rank': (_I, [_VP]),
    'adm_p2p_size': (_I, [_VP]),
    'adm_p2p_update': (_I, [_VP, _I, _VP, _VP, _SZ, _SZ, _SZ, _SZ, _I, _D, _D, _D, _D, _I, _VP]),
    'adm_p2p_all_reduce': (_I, [_VP, _VP, _SZ]),
    'adm_p2p_barrier': (_I, [_VP]),
    'adm_p2p_status': (_I, [_VP]),
    'adm_p2p_destroy': (_I, [_VP]),
    'adm_plan_create': (_I, [_VP, C.POINTER(PlanDesc), C.POINTER(_VP)]),
    'adm_plan_destroy': (_I, [_VP]),
    'adm_plan_set_detector_mask': (_I, [_VP, _VP]),
    'adm_plan_set_detector_kernels': (_I, [_VP, C.c_int, _VP, _VP]),
    'adm_plan_rot_elems': (_SZ, [_VP]),
    'adm_plan_workspace_bytes': (_SZ, [_VP, _I]),
    'adm_rotate_fwd': (_I, [_VP, _VP, _VP, _VP, _I, _I]),
    'adm_rotate_fwd_stack': (_I, [_VP, _VP, _VP, _I, _VP]),
    'adm_rotate_adj': (_I, [_VP, _VP, _VP, _VP, _I, _I]),
    'adm_rotate_adj_csr': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I]),
    'adm_rotate_adj_staged': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I]),
    'adm_rotate_adj_staged_stack': (_I, [_VP, _VP, _VP, _I, _VP, _VP, _SZ]),
    'adm_rotation_table_build': (_I, [_VP, _I, _I, C.c_float, C.c_float, _VP]),
    'adm_rotation_csr_scratch_bytes': (_SZ, [_VP]),
    'adm_rotation_csr_build': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _SZ]),
    'adm_multislice_fwd_adj': (_I, [_VP, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP, _VP, _F, _VP, _SZ]),
    'adm_plan_set_transmission_cache': (_I, [_VP, _I]),
    'adm_transmission_refresh': (_I, [_VP, _VP, _I, _I]),
    'adm_plan_set_generic': (_I, [_VP, _I]),
    'adm_multislice_fwd_adj_pp': (_I, [_VP, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP, _VP, _F, _VP, _SZ]),
    'adm_probe_shift': (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    'adm_probe_shift_adj': (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    'adm_center_rows': (_I, [_VP, _VP, _SZ, _I]),
    'adm_tile_grad_accumulate': (_I, [_VP, _VP, _SZ, _VP, _I, _VP, _VP]),
    'adm_tile_grad_accumulate_part': (_I, [_VP, _VP, _SZ, _VP, _I, _VP, _VP, _I, _I, _I]),
    'adm_tile_cover_build': (_I, [_VP, _VP, _SZ, _VP, _I, _VP, _I, _I, _I]),
    'adm_tile_grad_accumulate_range': (_I, [_VP, _VP, C.c_size_t, _VP, _I, _VP, _VP, _I, _I, _I]),
    'adm_tile_grad_status': (_I, [_VP, _VP, _SZ, _I, C.POINTER(_I)]),
    'adm_reg_grad': (_I, [_VP, _VP, _F, _F, _F, _VP, _VP]),
    'adm_reg_grad_range': (_I, [_VP, _VP, _F, _F, _F, _VP, _SZ, _SZ, _SZ, _SZ]),
    'adm_reg_grad_set': (_I, [_VP, _VP, _F, _F, _F, _VP, _VP]),
    'adm_adam_step': (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _SZ, _I, _D, _D, _D, _D, _I, _VP]),
    'adm_adam_step_small': (_I, [_VP, C.POINTER(SmallParam), _I, _I, _D, _D, _D]),
    'adm_gd_step': (_I, [_VP, _VP, _VP, _SZ, _SZ, _D, _I, _VP]),
    'adm_momentum_step': (_I, [_VP, _VP, _VP, _VP, _SZ, _SZ, _D, _D, _I, _VP]),
    'adm_rwl1_update': (_I, [_VP, _VP, _VP, _VP]),
    'adm_reg_grad_weighted': (_I, [_VP, _VP, _VP, _F, _F, _VP, _VP]),
    'adm_axpy': (_I, [_VP, _VP, _VP, _F, _SZ]),
    'adm_holo_create': (_I, [_VP, C.POINTER(HoloDesc), C.POINTER(_VP)]),
    'adm_holo_destroy': (_I, [_VP]),
    'adm_holo_fwd_adj': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    'adm_holo_fwd_adj_adam': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(HoloAdam), _VP, _VP]),
    'adm_holo_data_spectrum': (_I, [_VP, _VP, _VP]),
    'adm_holo_shift_targets': (_I, [_VP, _VP, _VP, _VP]),
    'adm_holo_set_registration': (_I, [_VP, _VP, C.c_int]),
    'adm_holo_shift_grad': (_I, [_VP, _VP, _VP, _VP, _VP]),
}

_lib = None


def load():
    """Load libadm.so (building nothing: run `python adorym_amd/csrc/build.py` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libadm.so is missing (%s). Build it with `python adorym_amd/csrc/build.py`; '
                           'adorym_amd has no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == ADM_OK:
        return
    msg = load().adm_last_error().decode('utf-8', 'replace')
    if rc == ADM_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ADM_ERR_INVALID:
        raise ValueError(msg)
    if rc == ADM_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError('libadm: %s (code %d)' % (msg, rc))
