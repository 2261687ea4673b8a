"""ctypes binding of libdmv.so (the C ABI declared in include/dmv.h).

The library is built in-tree by `make -C demiss_vae_amd/csrc` (see __graft_entry__.build) and is
required: there is no fallback path.  Loading fails loudly when it is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMV_LIB selects another build of the same library (e.g. the timestamp-instrumented tuning build)
_LIB_PATH = os.environ.get("DMV_LIB") or os.path.join(_HERE, "libdmv.so")
_lib = None

# enums (include/dmv.h)
ACT = {"relu": 0, "leaky_relu": 1, "tanh": 2, "sigmoid": 3}
IN_PLAIN, IN_MASKED, IN_CONCAT, IN_CONCAT_MASKED = 0, 1, 2, 3
GEN = {"normal": 0, "normal_with_eps": 1, "bern": 2, "bern_m1_p1": 3}
Q_OWN, Q_STRATIFIED, Q_IMPUTATION = 0, 1, 2
BOUND_LSE, BOUND_WEIGHTED, BOUND_PER_COMPONENT, TERMS_ONLY = 0, 1, 2, -1
E_BADARG, E_UNSUPPORTED, E_WORKSPACE = -1, -2, -3


class DmvDropRng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("offset", C.c_uint64), ("offset_base", C.c_void_p)]


DROP_GENERATED = C.c_void_p(1)      # dmv_mlp_backward's drop_keep when the forward drew its keep decisions itself


class DmvNet(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("hidden", C.c_int32), ("nb", C.c_int32),
                ("act", C.c_int32), ("dropout_p", C.c_float)]


class DmvObjCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("G", C.c_int32), ("nz", C.c_int32), ("I", C.c_int32),
                ("D", C.c_int32), ("Zd", C.c_int32), ("gen_dist", C.c_int32), ("q_eps", C.c_int32),
                ("x_per_component", C.c_int32),
                ("enc_sb", C.c_int64), ("enc_sg", C.c_int64), ("raw_off", C.c_int64), ("logit_off", C.c_int64)]


class DmvLatentBwd(C.Structure):
    _fields_ = [("base_prior", C.c_float), ("base_qown", C.c_float), ("base_qmix", C.c_float),
                ("base_w", C.c_float), ("kl_coef", C.c_float),
                ("own_live", C.c_int32), ("mix_live", C.c_int32), ("use_zscale", C.c_int32),
                ("mix_kind", C.c_int32)]


class DmvDecGrad(C.Structure):
    _fields_ = [("dec", C.c_void_p), ("ld_dec", C.c_int64), ("X", C.c_void_p), ("M", C.c_void_p), ("wgt", C.c_void_p),
                ("gb", C.c_void_p), ("base_m", C.c_float), ("base_f", C.c_float)]


class DmvObjFwd(C.Structure):
    _fields_ = [("q_kind", C.c_int32), ("mix_C", C.c_int32), ("kind", C.c_int32), ("use_logpi", C.c_int32), ("dreg", C.c_int32),
                ("coefA", C.c_float * 5), ("coefB", C.c_float * 5),
                ("terms", C.c_void_p), ("comp_lp", C.c_void_p), ("log_w", C.c_void_p),
                ("bound", C.c_void_p), ("mean_out", C.c_void_p), ("wgt", C.c_void_p), ("zscale", C.c_void_p), ("dlogit", C.c_void_p),
                ("objA", C.c_void_p), ("objB", C.c_void_p), ("meanA_out", C.c_void_p), ("meanB_out", C.c_void_p),
                ("neg_mean_out", C.c_void_p), ("ticket", C.c_void_p)]


class DmvPackMap(C.Structure):
    _fields_ = [("image", C.c_void_p), ("fwd_off", C.c_int64), ("tr_off", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
                ("S", C.c_int32), ("Sn", C.c_int32), ("is_bias", C.c_int32), ("reserved", C.c_int32)]


class DmvAdamGroup(C.Structure):
    _fields_ = [("lr", C.c_void_p), ("step", C.c_void_p), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_U64 = C.c_uint64
_F = C.c_float
_SZ = C.c_size_t

# name -> (restype, argtypes); every symbol include/dmv.h declares
SIGNATURES = {
    "dmv_version": (C.c_int, []),
    "dmv_strerror": (C.c_char_p, [C.c_int]),
    "dmv_pad32": (_I32, [_I32]),
    "dmv_mlp_packed_floats": (_SZ, [C.POINTER(DmvNet)]),
    "dmv_mlp_pack": (C.c_int, [C.POINTER(DmvNet), C.POINTER(_P), C.POINTER(_P), _P, _P]),
    "dmv_mlp_saved_floats": (_SZ, [C.POINTER(DmvNet), _I64]),
    "dmv_mlp_forward": (C.c_int, [C.POINTER(DmvNet), _P, _I32, _I32, _P, _I64, _P, _I64, _I64, _P, _I64, _P, _P,
                                  C.POINTER(DmvDropRng), _P]),
    "dmv_mlp_dropout_rng_elements": (_SZ, [C.POINTER(DmvNet), _I64]),
    "dmv_mlp_dropout_keep": (C.c_int, [C.POINTER(DmvNet), _I64, C.POINTER(DmvDropRng), _P, _P]),
    "dmv_mlp_backward_ws_bytes": (_SZ, [C.POINTER(DmvNet), _I64]),
    "dmv_mlp_backward": (C.c_int, [C.POINTER(DmvNet), _P, _P, _P, _P, _I64, _I64, _P, _I64,
                                   C.POINTER(_P), C.POINTER(_P), _F, _P, _SZ, _P]),
    "dmv_mlp_backward_wgrad": (C.c_int, [C.POINTER(DmvNet), _P, _P, _I64, _I64, _P, _P, _F, _P, _SZ, _P]),
    "dmv_mlp_backward_wgrad_pair": (C.c_int, [C.POINTER(DmvNet), _P, _P, _I64, _I64, _P, _P, _P, _SZ,
                                              C.POINTER(DmvNet), _P, _P, _I64, _I64, _P, _P, _P, _SZ, _F, _P]),
    "dmv_latent_sample": (C.c_int, [C.POINTER(DmvObjCfg), _P, _P, _U64, _U64, _P, _I64, _P, _P, _P, _P]),
    "dmv_vae_forward_fits": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(DmvNet), C.POINTER(DmvNet)]),
    "dmv_vae_forward": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(DmvNet), _P, _I32, _P, _P, C.POINTER(DmvNet), _P, _P, _U64, _U64,
                                  _P, _I32, _F, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "dmv_vae_forward_bound": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(DmvNet), _P, _I32, _P, _P, C.POINTER(DmvNet), _P, _P, _U64,
                                        _U64, _P, _P, _P, _P, _I64, _P, _P, _P, C.POINTER(DmvObjFwd), _P]),
    "dmv_lair_update_fits": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "dmv_lair_update": (C.c_int, [_P, _P, _I32, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, C.c_float, _I32, _P, _P, _P,
                                  _P, _P, _P, _U64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmv_pgibbs_update_fits": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32]),
    "dmv_pgibbs_update": (C.c_int, [_P, _P, _I32, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _U64, _P, _P, _P,
                                    _P, _P, _P, _P, _P]),
    "dmv_lair_resample_tail": (C.c_int, [_I32, _I32, _I32, _I32, _I32, _P, _P, _U64, _U64, _I32, _I32, _P, _I64, _P, _P, _P, _U64, _U64,
                                         _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmv_categorical_draw_grouped": (C.c_int, [_I32, _I32, _I32, _I32, _P, _P, _U64, _U64, _P, _P, _P, _I32, _P]),
    "dmv_vae_backward_fits": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(DmvNet), C.POINTER(DmvNet), _I32]),
    "dmv_vae_backward": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(DmvLatentBwd), C.POINTER(DmvDecGrad), C.POINTER(DmvNet), _P, _P, _P, _P, _I64, _P, _SZ,
                                   C.POINTER(DmvNet), _P, _P, _P, _P, _P, _SZ, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmv_obj_terms": (C.c_int, [C.POINTER(DmvObjCfg), _I32, _I32, _P, _I64, _P, _P, _P, _I64, _P, _P, _P, _P]),
    "dmv_obj_bound": (C.c_int, [C.POINTER(DmvObjCfg), _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmv_obj_combine": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(_F), _P, _P, _P, _P]),
    "dmv_kl_std_normal": (C.c_int, [C.POINTER(DmvObjCfg), _P, _P, _P]),
    "dmv_obj_forward": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(DmvObjFwd), _P, _I64, _P, _P, _P, _I64, _P, _P]),
    "dmv_obj_backward_dec": (C.c_int, [C.POINTER(DmvObjCfg), _P, _I64, _P, _P, _P, _P, _P, _F, _F, _P, _I64, _P]),
    "dmv_obj_backward_latent": (C.c_int, [C.POINTER(DmvObjCfg), C.POINTER(DmvLatentBwd), _P, _P, _P, _I64, _P, _I64,
                                          _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmv_obj_backward_dec_terms": (C.c_int, [C.POINTER(DmvObjCfg), _P, _I64, _P, _P, _P, _P, _P, _I64, _P]),
    "dmv_obj_backward_latent_terms": (C.c_int, [C.POINTER(DmvObjCfg), _I32, _I32, _P, _P, _P, _I64, _P, _I64,
                                                _P, _P, _P, _P, _P, _P, _P, _P]),
    "dmv_obj_implicit_ws_floats": (_SZ, [C.POINTER(DmvObjCfg), _I32]),
    "dmv_obj_backward_latent_implicit": (C.c_int, [C.POINTER(DmvObjCfg), _I32, _P, _P, _I64, _P, _I64, _P, _F, _F, _I32, _P, _P, _P]),
    "dmv_grid_marginal_ws_floats": (_SZ, [_I64, _I32]),
    "dmv_grid_marginal": (C.c_int, [_I32, _I32, _I64, _P, _I64, _P, _I32, _P, _P, _P, _P, _P, _P]),
    "dmv_opt_ws_floats": (_SZ, [_I32, _P]),
    "dmv_opt_adam_step": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _F, _P, _SZ, _P, _P, _P, _U64, _P]),
    "dmv_opt_adam_step_static": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _F, _P, _P, _P, _U64, _P]),
    "dmv_mlp_pack_map": (C.c_int, [C.POINTER(DmvNet), _P, C.POINTER(DmvPackMap), C.POINTER(DmvPackMap)]),
    "dmv_grid_divergences": (C.c_int, [_I32, _I32, _I32, _I64, _P, _P, _P, _F, _I32, _P, _P, _P, _I32, _P, _P, _P, _P]),
    "dmv_lse_update": (C.c_int, [_I32, _I64, _P, _I32, _P, _P, _P]),
    "dmv_rows_gather": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "dmv_rows_gather_multi": (C.c_int, [_I32, _P, _P, _P, _P, _I64, _P]),
    "dmv_rows_scatter": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "dmv_impute_tail": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _P, _P, _P, _U64, _U64, _P, _P, _P, _I32, _P, _P, _P, _P]),
    "dmv_categorical_draw": (C.c_int, [_I32, _I32, _I32, _P, _P, _U64, _U64, _P, _P, _P]),
    "dmv_lair_row_map": (C.c_int, [_I32, _I32, _I32, _P, _P, _P]),
    "dmv_dropout_masks": (C.c_int, [_P, _I64, _F, _U64, _U64, _P, _P]),
    "dmv_clock_watch": (C.c_int, [_I64, _P, _P]),
    "dmv_wall_clock_khz": (C.c_int, [C.POINTER(_I32)]),
}


def lib_path():
    return _LIB_PATH


def lib():
    """The loaded library.  Raises if libdmv.so has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C demiss_vae_amd/csrc)")
        l = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libdmv error {rc}: {lib().dmv_strerror(rc).decode()}")


_TICKETS = {}       # device index -> (zeroed int32 tensor of slots, {stream address: slot})


def ticket_ptr():
    """Address of the current stream's ticket counter for dmv_obj_forward (the launch then forms the batch means
    itself instead of launching a second kernel), or None when none can be had without allocating inside a stream
    capture.  Streams get a slot each, so launches on different streams never share a counter."""
    import torch
    dev = torch.cuda.current_device()
    stream = torch.cuda.current_stream().cuda_stream
    pool = _TICKETS.get(dev)
    if pool is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        pool = _TICKETS[dev] = (torch.zeros(64, dtype=torch.int32, device=f"cuda:{dev}"), {})
    slots, owner = pool
    slot = owner.get(stream)
    if slot is None:
        if len(owner) >= slots.numel():
            return None
        slot = owner[stream] = len(owner)
    return slots.data_ptr() + 4 * slot


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def addr(t):
    """Device address as an integer for pointer fields of ctypes structures (None -> NULL)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    arr = (_P * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
