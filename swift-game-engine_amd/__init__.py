"""MI355X (gfx950) character-update path for kelian343/swift-game-engine.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of
include/sge_amd.h), the ctypes mirror of that ABI (abi.py), the host driver
(engine.py, crowd.py), the CollisionQueryService mirror (services.py), the
shard-by-character exchange (parallel.py), the asset side (assets.py, formats.py,
fbx.py, exporters.py) and the reference-named C++ / Swift adapters (host/). The directory name carries a hyphen, so import it with
importlib.import_module("swift-game-engine_amd").
"""
from . import abi, assets, crowd, engine, exporters, fbx, formats, parallel, services  # noqa: F401
from .engine import CharacterEngine, SgeError, make_queries  # noqa: F401
