"""MI355X-native Differentiable Graph Generator hot path (package directory:
`learning-adaptive-neighborhoods-for-gnns_amd/`, importable as `dgg_amd` through the shim at the repo root).

Public surface mirrors the reference's modules:
    dgg_amd.dgm.{DGG_LearnableK_debug, DGG, LearnableKEncoder}               (reference dgm.py)
    dgg_amd.model.{GCNConv, GraphConvolution, DenseGraphConvolution, GCN_DGG, GCN_DGG_00, GCNII_DGG, GCNIIppi_DGG}  (model.py)
plus the containers `AllPairs` / `EllAdjacency` / `CsrAdjacency` and the raw kernel wrappers in `dgg_amd.ops`.
"""
from . import _lib, ops  # noqa: F401
from .adjacency import AllPairs, CsrAdjacency, EllAdjacency, csr_candidates, csr_pattern, ell_from_dense  # noqa: F401
from .dgm import (DGG, DGG_Ablations, DGG_LearnableK_debug, DGG_LearnableK_SDD, DGG_StraightThrough,  # noqa: F401
                  LearnableKEncoder)
from .model import (DenseGraphConv, DenseGraphConvolution, GAT_DGG_00, GAT_DGG_Ablations, GATConv_DGG, GCN_DGG, GCN_DGG_00,  # noqa: F401
                    GCN_DGG_Ablations, GCNConv, GCNII_DGG, GCNIIppi_DGG,
                    GraphConvolution, SAGE_DGG, SAGE_DGG_00)
