"""CPU-side checks of the drop-in boundary: C-ABI library exports, header <-> binding agreement,
state_dict layout / default init identical to the reference, checkpoint-prefix handling, loud failure
without a GPU.  No compute call is made here (there is no GPU in the build container)."""
import ctypes
import hashlib
import os

import numpy as np
import pytest
import torch

import awm_amd
from awm_amd import _lib
from oracle import recipes as R


def _sha(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 30
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/wm_hip.h but not exported"
    awm_amd.lib.load()


def test_header_types_are_plain_c():
    for name, args in _lib.parse_header().items():
        assert args[-1] == ("wm_stream_t", "stream"), name
        for ctype, _ in args:
            base = ctype.replace("const ", "").replace("*", "").strip()
            assert base in ("float", "int", "double", "long long", "void", "wm_stream_t"), (name, ctype)


def test_module_layout_and_init_match_reference(golden):
    torch.manual_seed(R.WEIGHT_SEED)
    G, D = awm_amd.Generator(message_bits=16), awm_amd.Detector(message_bits=16)
    assert np.array_equal(_sha(G.state_dict()), golden["init_sha_g"])      # same keys, order, shapes AND values
    assert np.array_equal(_sha(D.state_dict()), golden["init_sha_d"])
    assert len(G.state_dict()) == 53 and len(D.state_dict()) == 32
    assert G.message_bits == 16 and D.message_bits == 16
    assert not hasattr(awm_amd.Generator(0), "embedding")
    n_g = sum(p.numel() for p in G.parameters())
    n_d = sum(p.numel() for p in D.parameters())
    assert (n_g, n_d) == (4331777, 51537)
    # parameter order = the reference's (Adam param groups, py/main16.py:504)
    gsd, dsd = R.reference_layout_init()
    assert [k for k, _ in G.named_parameters()] == [k for k in gsd if "running" not in k and "num_batches" not in k]


def test_checkpoint_prefix_loader():
    ck = np.load(os.path.join(os.path.dirname(__file__), "golden", "detector_best_unprefixed.npz"))
    D = awm_amd.Detector(16)
    res = awm_amd.load_state_dict_strip_prefix(D, {"_orig_mod." + k: torch.from_numpy(ck[k]) for k in ck.files})
    assert not res.missing_keys and not res.unexpected_keys
    assert int(D.state_dict()["model.1.block.1.num_batches_tracked"]) == 4500
    res2 = awm_amd.load_state_dict_strip_prefix(awm_amd.Detector(16), {k: torch.from_numpy(ck[k]) for k in ck.files})
    assert not res2.missing_keys


def test_no_cpu_fallback():
    G = awm_amd.Generator(16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        G(torch.zeros(1, 1, 16000))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        awm_amd.postprocess(torch.zeros(1, 1, 16000))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        awm_amd.MultiScaleMelLoss()(torch.zeros(1, 1, 16000), torch.zeros(1, 1, 16000))


def test_product_does_not_import_oracle():
    pkg = os.path.dirname(awm_amd.__file__)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no oracle", ""), f"{fn} mentions the oracle"


def test_mel_tables_match_oracle_filterbank():
    from awm_amd import losses
    from oracle import wm_oracle as O
    fb, klo, khi, mlo = losses._mel_tables(torch.device("cpu"))
    assert torch.equal(fb, O.mel_filterbank())
    nz = fb > 0
    for m in range(64):
        ks = torch.nonzero(nz[:, m]).flatten()
        assert int(klo[m]) == int(ks[0]) and int(khi[m]) == int(ks[-1])
    assert torch.equal(losses._fir_taps(4000, 101, torch.device("cpu")), O.fir_kernel())
    assert int(np.floor(3500 * 512 / 16000)) + 1 == 113          # first penalised bin
