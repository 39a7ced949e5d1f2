"""MI355X-native Ising Monte Carlo sweep engine behind RRRMC.jl's graph / sampler API.

Host-side mirror (Python, since Julia is not available in this image) of the reference interface for
the hot path: ``Config`` (src/Interface.jl:21-29), ``GraphRRG`` (src/graphs/RRG.jl:116-162), ``GraphEA``
(src/graphs/EA.jl:138-193) and ``standardMC`` (src/RRRMC.jl:81-127).  All computation happens in the
gfx950 library ``lib/librrrmc_hip.so`` through the C ABI of ``include/rrrmc_hip.h``.
"""
from ._lib import RRRMCError, SYMBOLS, lib, pinned_empty  # noqa: F401
from .graphs import Config, GraphEA, GraphEANormal, GraphEANormalDiscretized, GraphQEAT, GraphQSKNormalT, GraphQSKT, GraphQuant, GraphRRG, GraphRRGNormal, GraphRRGNormalDiscretized, GraphSK, GraphSKNormal, all_delta_e, checkerboard_coloring, getN, level_units, neighbors  # noqa: F401
from .engine import EnergyProbe, Engine, bklMC, energy, extremal_opt, rrrMC, standardMC, wtmMC  # noqa: F401
from .observables import SnapshotLog, bitmatrix_chunks, get_ts_range, log_range, parseovs, parsets  # noqa: F401
from .sharding import gather_replica_major, shard_bounds  # noqa: F401

__all__ = ["Config", "GraphRRG", "GraphEA", "GraphSKNormal", "GraphSK", "GraphRRGNormal", "GraphEANormal", "GraphRRGNormalDiscretized", "GraphEANormalDiscretized", "GraphQuant", "GraphQSKT", "GraphQSKNormalT", "GraphQEAT", "rrrMC", "bklMC", "wtmMC", "extremal_opt", "checkerboard_coloring", "Engine", "EnergyProbe", "energy", "standardMC", "RRRMCError", "getN", "neighbors", "all_delta_e", "level_units",
           "shard_bounds", "gather_replica_major", "pinned_empty", "SnapshotLog", "parseovs", "parsets", "log_range", "get_ts_range", "bitmatrix_chunks"]
