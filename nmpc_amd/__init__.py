"""nmpc_amd: MI355X implementation of the sfwa/nmpc controller hot path.

The module-level functions below are the public interface of the reference's
Python module (python/nmpc/__init__.py:33-94), with the same names, argument
conventions (plain sequences in, list out of get_controls) and module globals
(HORIZON_LENGTH, STEP_LENGTH, state), bound to libcnmpc.so built from
nmpc_amd/csrc.  The batch interface for many instances is nmpc_amd.batch.Batch.

Differences from the reference module, all forced: the shared object is found
next to this file under its Linux name (the reference hard-codes .dylib paths,
__init__.py:99-103), and init() accepts and ignores the `implementation`
selector because there is one implementation.
"""
from ctypes import pointer

from . import _lib
from ._lib import NmpcError, State as _State  # noqa: F401

NMPC_PRECISION_FLOAT = 0
NMPC_PRECISION_DOUBLE = 1

_cnmpc = None
_REAL_T = None
_CONTROL_DIM = None
_STATE_DIM = None
state = None
HORIZON_LENGTH = 0
STEP_LENGTH = 0


def _need():
    if not _cnmpc:
        raise RuntimeError("Please call nmpc.init()")
    return _cnmpc


def integrate(dt, control=None):
    lib = _need()
    if control is None:
        control = (0.0, ) * _CONTROL_DIM
    elif len(control) != _CONTROL_DIM:
        raise ValueError("Control vector must contain %d elements" % _CONTROL_DIM)
    lib.nmpc_fixedwingdynamics_set_state(state)
    lib.nmpc_fixedwingdynamics_integrate(dt, (_REAL_T * _CONTROL_DIM)(*control))
    lib.nmpc_fixedwingdynamics_get_state(state)


def setup(state_weights, control_weights, terminal_weights,
          upper_control_bound, lower_control_bound):
    lib = _need()
    lib.nmpc_set_state_weights((_REAL_T * (_STATE_DIM - 1))(*state_weights))
    lib.nmpc_set_control_weights((_REAL_T * _CONTROL_DIM)(*control_weights))
    lib.nmpc_set_terminal_weights((_REAL_T * (_STATE_DIM - 1))(*terminal_weights))
    lib.nmpc_set_lower_control_bound((_REAL_T * _CONTROL_DIM)(*lower_control_bound))
    lib.nmpc_set_upper_control_bound((_REAL_T * _CONTROL_DIM)(*upper_control_bound))


def set_reference(point, index):
    _need().nmpc_set_reference_point(
        (_REAL_T * (_CONTROL_DIM + _STATE_DIM))(*point), index)


def initialise_horizon():
    _need().nmpc_init()


def prepare():
    _need().nmpc_preparation_step()


def set_rti(enabled=True):
    """Real-time-iteration mode (not in the reference module): prepare() then
    re-linearises the horizon about the last solution; see nmpc_set_rti."""
    _need().nmpc_set_rti(1 if enabled else 0)


def solve(measurement):
    _need().nmpc_feedback_step((_REAL_T * _STATE_DIM)(*measurement))


def get_controls():
    new_controls = (_REAL_T * _CONTROL_DIM)(*([0] * _CONTROL_DIM))
    _need().nmpc_get_controls(new_controls)
    return list(new_controls)


def get_result():
    """Controls plus the status nmpc_get_controls returns (0 = NMPC_OK,
    1 = NMPC_INFEASIBLE); the reference module drops the status
    (__init__.py:150-151)."""
    new_controls = (_REAL_T * _CONTROL_DIM)(*([0] * _CONTROL_DIM))
    status = _need().nmpc_get_controls(new_controls)
    return list(new_controls), status


def set_wind_velocity(wind_velocity):
    _need().nmpc_set_wind_velocity(
        _REAL_T(wind_velocity[0]), _REAL_T(wind_velocity[1]), _REAL_T(wind_velocity[2]))


def update_horizon(new_reference):
    _need().nmpc_update_horizon(
        (_REAL_T * (_STATE_DIM + _CONTROL_DIM))(*new_reference))


def init(implementation="c"):
    """Load libcnmpc.so and read its compiled configuration
    (python/nmpc/__init__.py:94-135)."""
    global _cnmpc, _REAL_T, _STATE_DIM, _CONTROL_DIM, state
    global HORIZON_LENGTH, STEP_LENGTH
    if implementation not in ("c", "c66x"):
        raise NameError(
            "Unknown NMPC implementation: %s (options are 'c', 'c66x')" % implementation)
    _cnmpc = _lib.load()
    # as the reference module does (__init__.py:123-124): the ctypes real type
    # follows the precision the library reports
    _REAL_T = _cnmpc.real
    _CONTROL_DIM = _cnmpc.nmpc_config_get_control_dim()
    _STATE_DIM = _cnmpc.nmpc_config_get_state_dim()
    HORIZON_LENGTH = _cnmpc.nmpc_config_get_horizon_length()
    STEP_LENGTH = _cnmpc.nmpc_config_get_step_length()
    state = _cnmpc.State()
    _cnmpc.nmpc_fixedwingdynamics_get_state(pointer(state))
