"""ctypes binding of libzigp.so (include/zigp.h).  No CPU fallback: a missing library is an error."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get('ZIGP_LIB') or os.path.join(ROOT, 'lib', 'libzigp.so')   # ZIGP_LIB: A/B builds on one GPU box

ZIGP_OK, ZIGP_EARG, ZIGP_EHIP, ZIGP_ENOTPD, ZIGP_ECOMM = 0, -1, -2, -3, -4
COMM_ID_BYTES = 128
NCLASS = 10
LIK_ONOFF, LIK_GAUSSIAN, LIK_BERNOULLI = 0, 1, 2   # include/zigp.h ZIGP_LIK_*
PROF_CLASSES = ('gemm_A1', 'gemm_A2', 'gemm_H', 'gemm_J', 'syrk', 'kuf_build', 'pointwise', 'kgrad', 'mxm_stage', 'other')
PROF_KERNELS = {'gemm_A1': 'gemm_f64_kernel<1,1,2,false,1,8,EpiStoreColsum> (A1 = W K, W read through W^T)',
                'gemm_A2': 'gemm_f64_kernel<1,1,2,false,2,8,EpiColsum> (A2 = W^T A1, reduced to the column sums sum_m s^2 A2^2 in the epilogue; the panel is not stored)',
                'gemm_H': 'gemm_f64_kernel<1,1,2,false,1,8,EpiStore> (H = W diag(s^2) A2; not launched since round 4: folded into J\')',
                'gemm_J': 'gemm_f64_kernel<1,1,2,false,0,8,EpiStorePanel> (J\' = Q A2 = (Q W^T) A1, Q = Kuu^-1 diag(s^2) - I: the products H = W diag(s^2) A2 and J\' = W^T H - A2 as one full product on the A1 panel)',
                'syrk': 'gemm_f64_kernel<0,0,2,true,3,4,EpiAccum> (C1 += A1 G A1^T)'}

# double* everywhere in the C-ABI; typed as void* on the Python side so that a plain address (ndarray.ctypes.data, 1 us) can be passed
# or stored in a struct field -- ndarray.ctypes.data_as(POINTER(c_double)) costs 2.5 us and a Kronecker step hands over ~30 arrays
dp = C.c_void_p


class ZigpError(RuntimeError):
    pass


class NotPositiveDefiniteError(ZigpError):
    """Cholesky failed (the reference raises tf.errors.InvalidArgumentError at this point)."""


class zigp_params(C.Structure):
    _fields_ = [('Mf', C.c_int32), ('Mg', C.c_int32), ('D', C.c_int32), ('reserved', C.c_int32),
                ('Zf', dp), ('Zg', dp), ('u_fm', dp), ('u_gm', dp), ('u_fs_sqrt', dp), ('u_gs_sqrt', dp),
                ('ell_f', dp), ('ell_g', dp), ('var_f', C.c_double), ('var_g', C.c_double), ('noise', C.c_double)]


class zigp_grads(C.Structure):
    _fields_ = [('Zf', dp), ('Zg', dp), ('u_fm', dp), ('u_gm', dp), ('u_fs_sqrt', dp), ('u_gs_sqrt', dp),
                ('ell_f', dp), ('ell_g', dp), ('var_f', C.c_double), ('var_g', C.c_double), ('noise', C.c_double)]


class zigp_kron_params(C.Structure):
    _fields_ = [('M0f', C.c_int32), ('M1f', C.c_int32), ('M0g', C.c_int32), ('M1g', C.c_int32),
                ('D0', C.c_int32), ('D1', C.c_int32), ('reserved0', C.c_int32), ('reserved1', C.c_int32),
                ('Z0f', dp), ('Z1f', dp), ('Z0g', dp), ('Z1g', dp),
                ('ell0f', dp), ('ell1f', dp), ('ell0g', dp), ('ell1g', dp),
                ('var0f', C.c_double), ('var1f', C.c_double), ('var0g', C.c_double), ('var1g', C.c_double),
                ('u_fm', dp), ('u_gm', dp), ('u_fs_sqrt', dp), ('u_gs_sqrt', dp), ('noise', C.c_double)]


class zigp_kron_grads(C.Structure):
    _fields_ = [('Z0f', dp), ('Z1f', dp), ('Z0g', dp), ('Z1g', dp),
                ('ell0f', dp), ('ell1f', dp), ('ell0g', dp), ('ell1g', dp),
                ('var0f', C.c_double), ('var1f', C.c_double), ('var0g', C.c_double), ('var1g', C.c_double),
                ('u_fm', dp), ('u_gm', dp), ('u_fs_sqrt', dp), ('u_gs_sqrt', dp), ('noise', C.c_double)]


FIT_BLOCKS = 17   # include/zigp.h ZIGP_FIT_BLOCKS


class zigp_kron_fit_opts(C.Structure):
    _fields_ = [('lr', C.c_double * FIT_BLOCKS), ('positive', C.c_int32 * FIT_BLOCKS), ('reserved', C.c_int32),
                ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double)]


# name -> (restype, argtypes); mirrors include/zigp.h and include/zigp_diag.h exactly (tests check every declared symbol resolves)
SIGNATURES = {
    'zigp_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    'zigp_destroy': (C.c_int, [C.c_void_p]),
    'zigp_last_error': (C.c_char_p, [C.c_void_p]),
    'zigp_last_info': (C.c_int, [C.c_void_p]),
    'zigp_set_chunk': (C.c_int, [C.c_void_p, C.c_int64]),
    'zigp_set_data': (C.c_int, [C.c_void_p, dp, dp, C.c_int64, C.c_int32]),
    'zigp_set_data_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    'zigp_select_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'zigp_elbo': (C.c_int, [C.c_void_p, C.POINTER(zigp_params), C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64,
                            C.c_int32, dp, dp, C.POINTER(zigp_grads)]),
    'zigp_predict': (C.c_int, [C.c_void_p, C.POINTER(zigp_params), dp, C.c_int64, C.c_double, C.c_double, dp]),
    'zigp_predict_device': (C.c_int, [C.c_void_p, C.POINTER(zigp_params), C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p]),
    'zigp_prior_kl': (C.c_int, [C.c_void_p, C.POINTER(zigp_params), C.c_double, dp]),
    'zigp_rbf_K': (C.c_int, [C.c_void_p, dp, C.c_int64, dp, C.c_int64, C.c_int32, dp, C.c_double, dp]),
    'zigp_kron_elbo': (C.c_int, [C.c_void_p, C.POINTER(zigp_kron_params), dp, dp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int32, dp, dp, C.POINTER(zigp_kron_grads), dp]),
    'zigp_kron_elbo_rows': (C.c_int, [C.c_void_p, C.POINTER(zigp_kron_params), C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_int32, dp, dp, C.POINTER(zigp_kron_grads), dp]),
    'zigp_kron_fit_steps': (C.c_int, [C.c_void_p, C.POINTER(zigp_kron_params), C.POINTER(zigp_kron_fit_opts), dp, dp, dp, C.c_int64, C.c_int64, C.c_int32,
                                      C.c_void_p, C.c_int64, dp, dp, C.c_double, C.c_double, C.c_int32, dp, dp]),
    'zigp_kron_fit_steps_applied': (C.c_int64, [C.c_void_p]),
    'zigp_test_trmm_list': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    'zigp_kron_predict': (C.c_int, [C.c_void_p, C.POINTER(zigp_kron_params), dp, C.c_int64, C.c_double, C.c_double, C.c_double, dp]),
    'zigp_get_chunk': (C.c_int64, [C.c_void_p, C.c_int32]),
    'zigp_get_chunk_rows': (C.c_int64, [C.c_void_p, C.c_int32, C.c_int64]),
    'zigp_set_pivot_rtol': (C.c_int, [C.c_void_p, C.c_double]),
    'zigp_comm_unique_id': (C.c_int, [C.c_void_p]),
    'zigp_comm_init': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'zigp_comm_destroy': (C.c_int, [C.c_void_p]),
    'zigp_comm_available': (C.c_int, [C.POINTER(C.c_int32)]),
    'zigp_comm_set_timeout': (C.c_int, [C.c_void_p, C.c_double]),
    'zigp_comm_allreduce_host': (C.c_int, [C.c_void_p, dp, C.c_int64]),
    'zigp_comm_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    'zigp_set_overlap': (C.c_int, [C.c_void_p, C.c_int32]),
    'zigp_set_kron_panels': (C.c_int, [C.c_void_p, C.c_int32]),
    'zigp_set_kron_range_tiles': (C.c_int, [C.c_void_p, C.c_int32]),
    'zigp_set_mean_function': (C.c_int, [C.c_void_p, dp, C.c_int32, C.c_double]),
    'zigp_get_mean_function_grad': (C.c_int, [C.c_void_p, dp, C.c_int32, dp]),
    'zigp_kron_head_elbo': (C.c_int, [C.c_void_p, C.POINTER(zigp_kron_params), C.c_int32, dp, dp, C.c_int64, C.c_double, C.c_double,
                                      C.c_double, C.c_int32, dp, dp, C.POINTER(zigp_kron_grads), dp]),
    'zigp_kron_head_predict': (C.c_int, [C.c_void_p, C.POINTER(zigp_kron_params), C.c_int32, dp, C.c_int64, C.c_double, C.c_double, dp]),
    'zigp_profile_enable': (C.c_int, [C.c_void_p, C.c_int32]),
    'zigp_profile_get': (C.c_int, [C.c_void_p, dp, C.POINTER(C.c_int64), dp]),
    'zigp_profile_reset': (C.c_int, [C.c_void_p]),
    'zigp_profile_totals': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'zigp_profile_sampling': (C.c_int, [C.c_void_p, C.c_int32]),
    'zigp_clock_stamp': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'zigp_test_gemm': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, dp, dp, dp]),
    'zigp_test_kuf': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, dp, dp, dp, C.c_double, dp]),
    'zigp_test_potrf_trtri': (C.c_int, [C.c_void_p, C.c_int64, dp, dp, dp, C.c_int32]),
}

_lib = None


def load():
    """Load libzigp.so; raises ZigpError when it has not been built (there is NO CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZigpError('libzigp.so not found at %s: build it first (python __graft_entry__.py / zigp.build.build()). '
                        'This engine has no CPU fallback.' % LIB_PATH)
    # PyTorch wheels bundle their own libamdhip64; a process must not end up with two HIP runtimes (the second one sees
    # no GPUs).  Loading torch's copy first lets libzigp.so bind to the same runtime, so engine buffers and torch tensors
    # (zigp_set_data_device, the all-reduce buffer) live in one context.  Without torch the system runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def as_f64(a, shape=None):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def ptr(a):
    """address of a float64 C-contiguous array (the caller keeps the array alive for the duration of the call)"""
    return a.ctypes.data
