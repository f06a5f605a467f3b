"""CPU-side checks of the reference-surface facade (no compute: there is no GPU here)."""
import pytest
import torch

import multi_modal_normative_modeling_amd as nm
from multi_modal_normative_modeling_amd import _lib
from tests.golden_util import Golden


def test_state_dict_interchange_with_reference_keys():
    g = Golden("mm3_gpoe")
    m = nm.cVAE_multimodal(g.dims, g.hidden, g.Z, g.c_dim, modalities=3, non_linear=True)
    w = g.weights("w0")
    assert list(m.state_dict().keys()) == list(w.keys())
    m.load_state_dict(w)
    sd = m.state_dict()
    assert all((sd[k] == v).all() for k, v in w.items())
    # the module tree exposes the reference's attribute names
    assert len(m.encoder_list) == 3 and len(m.decoder_list) == 3 and len(m.alpha_m_list) == 3
    assert hasattr(m, "optimizer1")
    assert m.encoder_list[1].encoder_layers[0].weight.shape == (g.hidden[0], g.dims[1] + g.c_dim)


def test_single_class_keys():
    g = Golden("single_small")
    m = nm.cVAE(g.dims[0], g.hidden, g.Z, g.c_dim, non_linear=True)
    assert list(m.state_dict().keys()) == list(g.weights("w0").keys())


def test_errors_match_reference_conventions():
    m = nm.cVAE_multimodal([5, 6], [8], 3, 2, modalities=2)
    with pytest.raises(ValueError, match="No such combination method"):        # cVAE.py:1163
        m.forward_multimodal([torch.zeros(2, 5), torch.zeros(2, 6)], [torch.zeros(2, 2)] * 2, "nope")
    if not torch.cuda.is_available():
        with pytest.raises(_lib.NmError):                                       # no CPU fallback
            m.forward_multimodal([torch.zeros(2, 5), torch.zeros(2, 6)], [torch.zeros(2, 2)] * 2, "poe")
    with pytest.raises(ValueError):
        nm.cVAE_multimodal([5, 6], [8], 3, 2, modalities=3)
