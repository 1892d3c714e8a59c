"""fusion_hip -- Python face of libfusion_hip.so, the MI355X implementation of the algebra
hot path of fusion-cryptography (batched negacyclic NTT/INTT, pointwise ring ops, the
small polynomial matrix-vector products and the fused keygen/sign/aggregate/verify cores).

``algebra`` and ``fusion`` (siblings of this package) mirror the reference's own modules on
top of it.  No CPU fallback exists: without the built library and a gfx950 device every
compute call raises FusionHipError.
"""
from ._lib import FusionHipError, LIB_PATH, SIGNATURES, load_library, runtime_report
from .context import (Comm, Context, DeviceArray, DeviceBuffer, Event, Graph, VERDICT_REASONS, comm_unique_id, get_context, rccl_library, rccl_version,
                      OP_ADD, OP_MUL, OP_NEG, OP_SUB)

__all__ = ["Comm", "Context", "DeviceArray", "DeviceBuffer", "Event", "FusionHipError", "Graph", "LIB_PATH", "SIGNATURES",
           "VERDICT_REASONS", "comm_unique_id", "get_context", "rccl_library", "rccl_version", "load_library", "runtime_report", "OP_ADD", "OP_MUL", "OP_NEG", "OP_SUB"]
