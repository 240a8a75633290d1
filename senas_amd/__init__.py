"""senas_amd -- MI355X-native implementation of the SENAS data-parallel hot path.

Package layout (only what the path needs):
  csrc/            HIP kernels + the C ABI (include/senas_hip.h) -> libsenas_hip.so
  _lib.py          ctypes binding of that ABI (no fallback: raises if the library is missing)
  functional.py    torch.autograd.Function wrappers (device pointers + current HIP stream)
  operations.py    OPS / OpType / candidate-op modules      (reference: utils/operations.py)
  cell.py          MixedOp, Cell                            (reference: search/cell.py)
  senas_search.py  SenasSearch, NAS, Architecture           (reference: search/senas_search.py)
  senas_model.py   BuildCell, SenasModel                    (reference: models/senas_model.py)
  genotype.py, geno_searched.py, loss.py, metrics.py, utils.py
"""
from .genotype import Genotype, GenoParser  # noqa: F401
from .operations import OPS, OpType, DownOps, UpOps, NormOps  # noqa: F401

__version__ = '0.1.0'
