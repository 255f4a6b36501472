"""cadnip.jl_amd -- MI355X-native transient inner loop for Cadnip.jl-style MNA simulation.

Host side (this package): device table, structure discovery, parameter packing and the
reference-shaped API (MNACircuit / dc / tran / CircuitSweep).  Device side: libcadnip_hip.so
(csrc/), reached only through the C ABI of include/cadnip_hip.h.
"""
from .circuit import Circuit, Param, Device  # noqa: F401
from .structure import discover, pack_params, expand_breakpoints, Structure  # noqa: F401
from . import mos1_params, bsource, netlist  # noqa: F401

__all__ = ["Circuit", "Param", "Device", "discover", "pack_params", "expand_breakpoints", "Structure"]
