"""Host-side behaviour of the fused-head module that needs no GPU: `install_fused_head` leaves the model's ordinary
forward and its parameter names alone, refuses models without the body / head split, and the loss path fails
loudly (no CPU fallback) when asked to run off-device."""

from __future__ import annotations

import types

import pytest
import torch

from pipelinerl_amd.finetune.rl import RLConfig
from pipelinerl_amd.finetune.types import PipelineBatchEncoding
from pipelinerl_amd.fused_head import FusedLmHead, install_fused_head, rl_step_fused_head

V, H, T = 128, 64, 12


class _Body(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.emb = torch.nn.Embedding(V, H)

    def forward(self, input_ids=None, **kw):
        return (self.emb(input_ids).to(torch.bfloat16),)


class _LM(torch.nn.Module):
    def __init__(self, bias: bool = False):
        super().__init__()
        self.model = _Body()
        self.lm_head = torch.nn.Linear(H, V, bias=bias)

    def forward(self, input_ids=None, **kw):
        return types.SimpleNamespace(logits=self.lm_head(self.model(input_ids=input_ids)[0].float()))


def _batch():
    ids = torch.randint(0, V, (1, T))
    z = torch.zeros(1, T)
    return PipelineBatchEncoding(input_ids=ids, labels=ids.clone(), position_ids=torch.arange(T)[None], attention_mask=torch.ones(1, T, dtype=torch.int64),
                                 old_logprobs=z, ref_logprobs=z, advantages=z, rewards=z, group_tokens=z + 1, num_labels=z + T, overflow=z,
                                 model_version=0, is_packed=True)


def test_install_keeps_forward_and_names():
    torch.manual_seed(0)
    lm = _LM()
    ids = torch.randint(0, V, (1, T))
    before = lm(input_ids=ids).logits
    names = [n for n, _ in lm.named_parameters()]
    assert install_fused_head(lm, chunk_rows=256, hidden_grad_terms=1) is lm
    assert install_fused_head(lm) is lm  # idempotent: the first options stay
    assert lm._prl_fused_head == {"chunk_rows": 256, "hidden_grad_terms": 1, "keep_logits": None}
    assert torch.equal(lm(input_ids=ids).logits, before)
    assert [n for n, _ in lm.named_parameters()] == names
    assert set(lm.state_dict()) == set(names)


def test_models_without_a_separate_bias_free_head_are_refused():
    with pytest.raises(TypeError):
        install_fused_head(torch.nn.Linear(4, 4))
    with pytest.raises(TypeError):
        install_fused_head(_LM(bias=True))
    with pytest.raises(TypeError):
        rl_step_fused_head(torch.nn.Linear(4, 4), _batch(), 0, 1, RLConfig())


def test_no_cpu_fallback():
    lm = install_fused_head(_LM())
    with pytest.raises(Exception) as e:  # the HIP path refuses host tensors instead of computing something else
        rl_step_fused_head(lm, _batch(), 0, 1, RLConfig())
    assert "device" in str(e.value).lower() or "cuda" in str(e.value).lower() or "hip" in str(e.value).lower()
    with pytest.raises(Exception) as e:  # gspo is served by the fused head too - and refuses host tensors like the rest
        rl_step_fused_head(lm, _batch(), 0, 1, RLConfig(policy_loss="gspo"))
    assert not isinstance(e.value, NotImplementedError)
    with pytest.raises(ValueError):
        FusedLmHead(torch.zeros(4))
    with pytest.raises(TypeError):
        FusedLmHead(torch.zeros(4, 4, dtype=torch.float16))


@pytest.mark.parametrize("config,reason", [({"final_logit_softcapping": 30.0}, "softcapping"), ({"logit_scale": 0.0625}, "logit_scale"),
                                           ({"logits_scaling": 8.0}, "logits_scaling"), ({"vocab_size": V - 8}, "rows")])
def test_models_that_postprocess_their_logits_are_not_given_the_fused_head(config, reason):
    """hidden @ W^T is not the model's `.logits` for an architecture with soft-capping, a logit scale or a padded head: the training
    path refuses (TypeError naming the attribute), the reference-policy path quietly keeps using the model's own logits."""
    from pipelinerl_amd.fused_head import _logits_postprocessing, ref_head_for

    lm = _LM()
    lm.config = types.SimpleNamespace(vocab_size=V, **config) if "vocab_size" not in config else types.SimpleNamespace(**config)
    assert reason in _logits_postprocessing(lm)
    with pytest.raises(TypeError, match="post-processes its logits"):
        install_fused_head(lm)
    assert ref_head_for(lm) is None
    # the neutral values of the same attributes are fine
    ok = _LM()
    ok.config = types.SimpleNamespace(vocab_size=V, final_logit_softcapping=None, logit_scale=1.0)
    assert _logits_postprocessing(ok) is None and install_fused_head(ok) is ok
