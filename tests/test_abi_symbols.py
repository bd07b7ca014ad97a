"""CPU tests of the drop-in boundary: libcnmpc.so loads without a GPU and
exports every function include/cnmpc.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cnmpc.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(nmpc_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_t")))


# the 24 entry points of the reference's c/cnmpc.h:52-99
REFERENCE_ABI = """nmpc_init nmpc_preparation_step nmpc_feedback_step nmpc_get_controls
nmpc_update_horizon nmpc_set_state_weights nmpc_set_control_weights
nmpc_set_terminal_weights nmpc_set_lower_control_bound nmpc_set_upper_control_bound
nmpc_set_reference_point nmpc_set_wind_velocity nmpc_fixedwingdynamics_set_position
nmpc_fixedwingdynamics_set_velocity nmpc_fixedwingdynamics_set_attitude
nmpc_fixedwingdynamics_set_angular_velocity nmpc_fixedwingdynamics_set_state
nmpc_fixedwingdynamics_get_state nmpc_fixedwingdynamics_integrate
nmpc_config_get_state_dim nmpc_config_get_control_dim nmpc_config_get_horizon_length
nmpc_config_get_step_length nmpc_config_get_precision""".split()


def test_header_declares_reference_abi():
    decl = declared_functions()
    assert len(REFERENCE_ABI) == 24
    for name in REFERENCE_ABI:
        assert name in decl, name


def test_library_exports_every_declared_symbol():
    from nmpc_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `make -C nmpc_amd/csrc`"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), "missing export: " + name


def test_prototype_table_matches_header():
    from nmpc_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_functions()


def test_compiled_configuration():
    """Config getters need no GPU (ccs-c66x/cnmpc.c:1088-1110)."""
    from nmpc_amd import _lib
    lib = _lib.load()
    assert lib.nmpc_config_get_state_dim() == 13
    assert lib.nmpc_config_get_control_dim() == 3
    assert lib.nmpc_config_get_horizon_length() == 100
    assert lib.nmpc_config_get_step_length() == 1.0 / 50.0
    assert lib.nmpc_config_get_precision() == 1     # NMPC_PRECISION_DOUBLE
    assert ctypes.sizeof(_lib.State) == 13 * 8


def test_single_precision_library_exports_and_configuration():
    """libcnmpc_f32.so: the same entry points with real_t = float, the
    reference's shipped configuration (include/config.h:27)."""
    from nmpc_amd import _lib
    path = os.path.join(ROOT, "nmpc_amd", "libcnmpc_f32.so")
    assert os.path.exists(path), "build with `make -C nmpc_amd/csrc`"
    raw = ctypes.CDLL(path)
    for name in declared_functions():
        assert hasattr(raw, name), "missing export: " + name
    lib = _lib.load(path)
    assert lib.nmpc_config_get_precision() == 0     # NMPC_PRECISION_FLOAT
    assert lib.real is ctypes.c_float and lib.np_real == "float32"
    assert ctypes.sizeof(lib.State) == 13 * 4
    assert lib.nmpc_config_get_step_length() == 1.0 / 50.0 or \
        lib.nmpc_config_get_step_length() == float(__import__("numpy").float32(1.0 / 50.0))


def test_missing_library_fails_loudly(tmp_path):
    from nmpc_amd import _lib
    with pytest.raises(_lib.NmpcError):
        _lib.load(str(tmp_path / "libcnmpc.so"))


def test_batch_argument_errors_without_gpu():
    """Argument validation happens before any HIP call."""
    from nmpc_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.nmpc_batch_create(ctypes.byref(h), 0, 100, 0, None) == -1   # NMPC_E_ARG
    assert b"instances" in lib.nmpc_last_error()
    assert lib.nmpc_batch_create(None, 1, 100, 0, None) == -1
    # the kernels index (instance, stage) pairs with 32-bit integers
    assert lib.nmpc_batch_create(ctypes.byref(h), 1 << 24, 127, 0, None) == -1
    assert b"2^31" in lib.nmpc_last_error()
    assert lib.nmpc_batch_feedback_step(None, None) == -1


def test_python_module_surface():
    """Same public names as the reference's python/nmpc/__init__.py:33-94."""
    import nmpc_amd
    for name in ("integrate", "setup", "set_reference", "initialise_horizon", "prepare",
                 "solve", "get_controls", "set_wind_velocity", "update_horizon", "init"):
        assert callable(getattr(nmpc_amd, name)), name
    if nmpc_amd._cnmpc is None:
        with pytest.raises(RuntimeError):
            nmpc_amd.prepare()
    with pytest.raises(NameError):
        nmpc_amd.init("fortran")
