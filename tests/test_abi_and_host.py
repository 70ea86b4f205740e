"""CPU-side checks: the C-ABI library loads and exports every symbol include/dgg_hip.h declares (no compute
calls without a GPU); the drop-in modules keep the reference's state_dict contract; ops refuse CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, load_fixture


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dgg_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dgg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import dgg_amd
    syms = declared_symbols()
    assert len(syms) >= 19
    L = ctypes.CDLL(dgg_amd._lib.SO_PATH)
    for s in syms:
        assert hasattr(L, s), f"libdgg_hip.so does not export {s}"
    # the ctypes prototype table covers the same set (minus the two info functions)
    assert set(dgg_amd._lib.PROTOTYPES) | {"dgg_last_error", "dgg_abi_version"} == set(syms)
    assert dgg_amd._lib.lib().dgg_abi_version() == 5


def test_public_header_is_valid_c():
    """include/dgg_hip.h must compile as plain C (it is what a cgo / JNI / ctypes binding generator would read)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "include", "dgg_hip.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_state_dict_contract_matches_reference_fixture():
    """keys/shapes of the reference's DGG_LearnableK_debug.state_dict() (captured in the golden fixture)"""
    import dgg_amd
    from argparse import Namespace
    fx = load_fixture("allpairs_n256_asym")
    args = Namespace(**fx["meta"]["args"])
    m = dgg_amd.DGG_LearnableK_debug(in_dim=fx["meta"]["d"], latent_dim=fx["meta"]["h"], args=args)
    ref = {k[2:]: v.shape for k, v in fx.items() if k.startswith("p.")}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == {k: tuple(s) for k, s in ref.items()}
    m.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("p.")}, strict=True)


def test_model_wrappers_keys():
    import dgg_amd
    from argparse import Namespace
    args = Namespace(extra_edge_dim=2, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288,
                     dgg_mode_edge_net="u-v-dist", dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob",
                     debug_step=3, perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False,
                     dgg_adj_input="input_adj", n_dgg_layers=1)
    m = dgg_amd.GCN_DGG(nfeat=1433, nlayers=2, nhidden=64, nclass=7, args=args)
    sd = m.state_dict()
    assert len(sd) == 38                                   # SURVEY.md section 8b [probe]
    assert sum(p.numel() for p in m.parameters()) == 300257
    for k in ("convs.0.W", "convs.1.W", "conv1.W", "conv2.W", "dggs.0.t", "dggs.0.k_W",
              "dggs.0.node_encode_for_edges.0.weight", "dggs.0.k_net.k_project.bias"):
        assert k in sd
    g = dgg_amd.GCNII_DGG(nfeat=50, nlayers=4, nhidden=32, nclass=3, dropout=0.5, lamda=0.5, alpha=0.1, variant=False, args=args)
    ks = g.state_dict().keys()
    assert "convs.3.weight" in ks and "fcs.0.weight" in ks and "fcs.1.bias" in ks and "dggs.0.k_embed.0.weight" in ks


def test_ops_refuse_cpu_tensors_and_unsupported_modes():
    import dgg_amd
    from argparse import Namespace
    with pytest.raises(AssertionError):
        dgg_amd.ops.linear_fwd(torch.zeros(4, 4), torch.zeros(4, 4))
    args = Namespace(extra_edge_dim=2, extra_k_dim=1, dgg_hard=False, deg_mean=3.899, deg_std=5.288,
                     dgg_mode_edge_net="nonsense", dgg_mode_k_net="x", dgg_mode_k_select="k_times_edge_prob",
                     debug_step=3, perturb_edge_prob=True, symmetric_noise=False, stochastic_k=False)
    m = dgg_amd.DGG_LearnableK_debug(8, 16, args)
    with pytest.raises(Exception, match="mode not found"):      # reference dgm.py:1726-1727
        m(torch.zeros(4, 8), dgg_amd.AllPairs(torch.ones(4)))


def test_missing_library_fails_loudly(tmp_path):
    """no CPU fallback: importing the package without libdgg_hip.so must raise as soon as a kernel is requested"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import dgg_amd; "
            "\ntry:\n    dgg_amd._lib.lib()\n    print('LOADED')\nexcept dgg_amd._lib.DggHipError as e:\n    print('RAISED', 'no CPU fallback' in str(e))") % ROOT
    env = dict(os.environ, DGG_HIP_SO=str(tmp_path / "nope" / "libdgg_hip.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "RAISED True" in r.stdout, r.stdout + r.stderr
