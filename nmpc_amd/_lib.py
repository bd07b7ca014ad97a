"""ctypes loader for nmpc_amd/libcnmpc.so (prototypes of include/cnmpc.h).

There is no CPU fallback: a missing library is an ImportError-like failure at
first use, and every batch call raises NmpcError on a negative return code.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NMPC_AMD_LIB names another build of the same library (the parity tests load the
# build whose factorisation always takes its plain-division path this way).
LIB_PATH = os.environ.get("NMPC_AMD_LIB") or os.path.join(HERE, "libcnmpc.so")

REAL = C.c_double
PREAL = C.POINTER(C.c_double)
PI32 = C.POINTER(C.c_int32)

K_LINEARISE, K_QP_BEGIN, K_LINE_SEARCH_TAIL, K_NEWTON_SOLVE, K_NEWTON_STEP, K_OTHER = range(6)
KERNEL_NAMES = ("linearise", "qp_begin", "line_search_tail", "newton_solve",
                "newton_step", "other")


class NmpcError(RuntimeError):
    pass


_state_types = {}


def state_type(real):
    """struct nmpc_state_t (c/cnmpc.h:39-44) for a library whose real_t is `real`."""
    if real not in _state_types:
        class State(C.Structure):
            _fields_ = [("position", real * 3), ("velocity", real * 3),
                        ("attitude", real * 4), ("angular_velocity", real * 3)]

            def __repr__(self):
                return str({"position": tuple(self.position), "velocity": tuple(self.velocity),
                            "attitude": tuple(self.attitude),
                            "angular_velocity": tuple(self.angular_velocity)})
        _state_types[real] = State
    return _state_types[real]


State = state_type(C.c_double)


# name -> (restype, argtypes); the exported surface of include/cnmpc.h
def prototypes(REAL):
    """name -> (restype, argtypes) for a library whose real_t is REAL."""
    PREAL = C.POINTER(REAL)
    State = state_type(REAL)
    return {
        # reference controller interface
        "nmpc_init": (None, []),
        "nmpc_preparation_step": (None, []),
        "nmpc_feedback_step": (None, [PREAL]),
        "nmpc_get_controls": (C.c_int, [PREAL]),
        "nmpc_update_horizon": (None, [PREAL]),
        "nmpc_set_state_weights": (None, [PREAL]),
        "nmpc_set_control_weights": (None, [PREAL]),
        "nmpc_set_terminal_weights": (None, [PREAL]),
        "nmpc_set_lower_control_bound": (None, [PREAL]),
        "nmpc_set_upper_control_bound": (None, [PREAL]),
        "nmpc_set_reference_point": (None, [PREAL, C.c_uint32]),
        "nmpc_set_wind_velocity": (None, [REAL, REAL, REAL]),
        "nmpc_fixedwingdynamics_set_position": (None, [REAL, REAL, REAL]),
        "nmpc_fixedwingdynamics_set_velocity": (None, [REAL, REAL, REAL]),
        "nmpc_fixedwingdynamics_set_attitude": (None, [REAL, REAL, REAL, REAL]),
        "nmpc_fixedwingdynamics_set_angular_velocity": (None, [REAL, REAL, REAL]),
        "nmpc_fixedwingdynamics_set_state": (None, [C.POINTER(State)]),
        "nmpc_fixedwingdynamics_get_state": (None, [C.POINTER(State)]),
        "nmpc_fixedwingdynamics_integrate": (None, [C.c_float, PREAL]),
        "nmpc_config_get_state_dim": (C.c_uint32, []),
        "nmpc_config_get_control_dim": (C.c_uint32, []),
        "nmpc_config_get_horizon_length": (C.c_uint32, []),
        "nmpc_config_get_step_length": (REAL, []),
        "nmpc_config_get_precision": (C.c_int, []),
        # batch interface
        "nmpc_last_error": (C.c_char_p, []),
        "nmpc_batch_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32,
                                        C.c_int, C.c_void_p]),
        "nmpc_batch_destroy": (C.c_int, [C.c_void_p]),
        "nmpc_batch_synchronize": (C.c_int, [C.c_void_p]),
        "nmpc_batch_reset": (C.c_int, [C.c_void_p]),
        "nmpc_batch_set_state_weights": (C.c_int, [C.c_void_p, PREAL]),
        "nmpc_batch_set_control_weights": (C.c_int, [C.c_void_p, PREAL]),
        "nmpc_batch_set_terminal_weights": (C.c_int, [C.c_void_p, PREAL]),
        "nmpc_batch_set_lower_control_bound": (C.c_int, [C.c_void_p, PREAL]),
        "nmpc_batch_set_upper_control_bound": (C.c_int, [C.c_void_p, PREAL]),
        "nmpc_batch_set_instance_weights": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                      C.c_void_p, C.c_void_p]),
        "nmpc_batch_set_instance_control_bounds": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                             C.c_void_p]),
        "nmpc_batch_init": (C.c_int, [C.c_void_p]),
        "nmpc_batch_set_wind_velocity": (C.c_int, [C.c_void_p, C.c_void_p]),
        "nmpc_batch_set_reference_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32,
                                                      C.c_uint32]),
        "nmpc_batch_set_reference_from_waypoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, REAL,
                                                              REAL, C.c_uint32, C.c_uint32]),
        "nmpc_batch_set_reference_from_waypoints_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32,
                                                                      REAL, REAL, C.c_uint32, C.c_uint32]),
        "nmpc_batch_preparation_step": (C.c_int, [C.c_void_p]),
        "nmpc_batch_set_rti": (C.c_int, [C.c_void_p, C.c_int]),
        "nmpc_set_rti": (None, [C.c_int]),
        "nmpc_batch_relinearise": (C.c_int, [C.c_void_p]),
        "nmpc_batch_feedback_step": (C.c_int, [C.c_void_p, C.c_void_p]),
        "nmpc_batch_get_controls": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "nmpc_batch_update_horizon": (C.c_int, [C.c_void_p, C.c_void_p]),
        "nmpc_batch_plant_set_state": (C.c_int, [C.c_void_p, C.c_void_p]),
        "nmpc_batch_plant_get_state": (C.c_int, [C.c_void_p, C.c_void_p]),
        "nmpc_batch_plant_integrate": (C.c_int, [C.c_void_p, REAL, C.c_void_p, C.c_void_p]),
        "nmpc_batch_feedback_step_from_plant": (C.c_int, [C.c_void_p]),
        "nmpc_batch_instances": (C.c_uint32, [C.c_void_p]),
        "nmpc_batch_horizon": (C.c_uint32, [C.c_void_p]),
        "nmpc_batch_get_stage": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, PREAL]),
        "nmpc_batch_global_size": (C.c_uint32, [C.c_void_p]),
        "nmpc_batch_get_global": (C.c_int, [C.c_void_p, C.c_uint32, PREAL]),
        "nmpc_batch_get_solver_info": (C.c_int, [C.c_void_p, C.c_uint32, PI32, PREAL]),
        "nmpc_batch_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
        "nmpc_batch_set_profiling_kinds": (C.c_int, [C.c_void_p, C.c_uint]),
        "nmpc_batch_get_kernel_time": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double),
                                                 C.POINTER(C.c_uint64)]),
        "nmpc_linearise_samples": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64] +
                                   [C.c_void_p] * 4 + [PREAL, PREAL] + [C.c_void_p] * 2),
        "nmpc_rk4_samples": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64] + [C.c_void_p] * 3 +
                             [REAL, C.c_void_p]),
        "nmpc_measure_fp64_vector_peak": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double)]),
        "nmpc_measure_stream_triad": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_double)]),
    }



PROTOTYPES = prototypes(C.c_double)      # the names, and the types of the double build

_lib = None


def _load_torch_first():
    """PyTorch-ROCm wheels bundle their own HIP runtime under the same soname
    as the system one (libamdhip64.so.7).  The copy loaded first serves every
    later user; if libcnmpc.so pulled in the system copy first, a later
    `import torch` would bring a second runtime into the process and fail with
    "No HIP GPUs are available".  So torch, when installed, goes first.
    Set NMPC_AMD_NO_TORCH=1 to skip this in torch-free deployments."""
    if os.environ.get("NMPC_AMD_NO_TORCH") == "1":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules or importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load(path=None):
    """Load the shared library and attach prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise NmpcError(
            "%s not found: build it with `make -C nmpc_amd/csrc` (hipcc, gfx950); "
            "there is no CPU fallback" % p)
    _load_torch_first()
    lib = C.CDLL(p)
    # real_t of this build: double, or float for the single-precision library
    lib.nmpc_config_get_precision.restype = C.c_int
    single = lib.nmpc_config_get_precision() == 0       # NMPC_PRECISION_FLOAT
    lib.real = C.c_float if single else C.c_double
    lib.preal = C.POINTER(lib.real)
    lib.np_real = "float32" if single else "float64"
    lib.State = state_type(lib.real)
    for name, (res, args) in prototypes(lib.real).items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(lib, rc):
    if rc != 0:
        raise NmpcError("libcnmpc error %d: %s" % (rc, lib.nmpc_last_error().decode()))
