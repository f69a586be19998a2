"""lap-time-optimization_amd — MI355X-native (gfx950) receding-horizon NLP solver behind the reference's
`controller.mpc.make_step(x0) -> u0` surface (bruno-maruszczak/lap-time-optimization, src/mpc.py:142).

The directory name contains a hyphen; import it with importlib.import_module("lap-time-optimization_amd")
or through the alias module `ltompc` at the repository root.
"""
from .tables import TrackTables, build_tables  # noqa: F401
from ._lib import LtompcError, Options, Params, default_options, default_params, STATUS_NAMES, NO_BOUND  # noqa: F401
from .solver import BatchedMPC, SplitMPC  # noqa: F401
from .mpc import Controller, Simulator, Track, VehicleModel, closed_loop  # noqa: F401
from .scenarios import X0_REFERENCE, sample_x0  # noqa: F401
from .velocity import Vehicle, VehicleMX5, VelocityProfile, VpVehicle  # noqa: F401
from ._build import build as build_library  # noqa: F401
