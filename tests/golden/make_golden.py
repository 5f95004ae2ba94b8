"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Run in the build container (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py

It imports `pipelinerl.finetune.{rl, data, utils}` from /root/reference (stubbing the absent
`omegaconf`, which the reference only uses in annotations), feeds them the seeded synthetic
rollouts of `pipelinerl_amd.synthetic`, and stores inputs + reference outputs:

    preprocess_*.json / .npz   raw rollouts -> preprocess_fn + populate_rl_data scalars,
                               collate_packed / collate tensors
    rl_step_*.npz              batch + logits -> rl_step loss, 32 stats, autograd d loss/d logits
    sentinel.npz               create_sentinel_batch fields

The reference has no tests or golden vectors of its own for this path; these
files are what pins `oracle/` and, through it, the HIP kernels.
"""

from __future__ import annotations

import copy
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REFERENCE = Path("/root/reference")

sys.path.insert(0, str(REPO))


def import_reference():
    if not REFERENCE.exists():
        raise SystemExit("/root/reference not found: golden fixtures can only be regenerated in the build container")
    sys.path.insert(0, str(REFERENCE))
    om = types.ModuleType("omegaconf")
    for name in ("DictConfig", "ListConfig", "OmegaConf"):
        setattr(om, name, type(name, (), {}))
    sys.modules.setdefault("omegaconf", om)
    import pipelinerl.finetune.data as ref_data
    import pipelinerl.finetune.rl as ref_rl
    import pipelinerl.finetune.utils as ref_utils

    return ref_rl, ref_data, ref_utils


class Tok:
    """All that collate / populate touch on a tokenizer."""

    def __init__(self, eos_token_id=2, padding_side="right"):
        self.eos_token_id = eos_token_id
        self.padding_side = padding_side


EOS = 2


def ref_preprocess(ref_rl, ref_data, raw, rl_config):
    """preprocess_dataset minus OOV patching / reference LLM (preprocess.py:145-189)."""
    tok = Tok(EOS)
    data = copy.deepcopy(raw)
    for e in data:
        if not e.get("ref_logprobs"):
            e["ref_logprobs"] = e["logprobs"]
    dataset = []
    for e in data:
        entry = dict(e)
        entry.update(ref_data.preprocess_fn(e, tokenizer=tok, seq_length=10**6, is_rl=True))
        dataset.append(entry)
    for entry in dataset:
        entry["model_version"] = entry["metadata"]["model_version"]
        entry["rollout_index"] = entry["metadata"]["rollout_index"]
        entry["step_index"] = entry["metadata"]["step_index"]
    return ref_rl.populate_rl_data(dataset=dataset, eos_token_id=EOS, config=rl_config)


def batch_to_np(b):
    out = {}
    for k, v in b.model_dump().items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        elif v is not None:
            out[k] = np.asarray(v)
    return out


def make_raw_cases():
    from pipelinerl_amd.synthetic import make_entries

    cases = {}
    # p1: plain groups of 4, binary rewards
    cases["p1"] = dict(raw=make_entries(3, attempts=4, seq_length=96, vocab=500, seed=11, prompt_min=8, prompt_max=24),
                       divide_advantage_by_std=False)
    # p2: std-normalised, explicit ref logprobs, float rewards, multi-step rollouts, a singleton group,
    #     an unfinished rollout that nevertheless contains EOS (EOS-scan branch)
    raw = make_entries(3, attempts=4, seq_length=80, vocab=300, seed=12, prompt_min=6, prompt_max=20, with_ref=True)
    rng = np.random.default_rng(5)
    for e in raw:
        e["reward"] = float(np.round(rng.uniform(-1, 2), 3))
    for k, e in enumerate(raw[:4]):  # group g0: 2 rollouts x 2 steps
        e["metadata"]["rollout_index"] = k // 2
        e["metadata"]["step_index"] = k % 2
    raw[-1]["group_id"] = "solo"  # singleton group
    for e in raw:
        if not e["finished"] and "finish_reason" not in e:
            e["input_ids"][1] = EOS
            break
    cases["p2"] = dict(raw=raw, divide_advantage_by_std=True)
    # p3: dense, all the same reward in one group (zero variance), attempts 8
    raw = make_entries(2, attempts=8, seq_length=64, vocab=200, seed=13, prompt_min=4, prompt_max=12, dense=True)
    for e in raw[:8]:
        e["reward"] = 1.0
    cases["p3"] = dict(raw=raw, divide_advantage_by_std=True)
    return cases


def gen_preprocess(ref_rl, ref_data):
    prepared = {}
    for name, case in make_raw_cases().items():
        cfg = ref_rl.RLConfig(divide_advantage_by_std=case["divide_advantage_by_std"])
        data = ref_preprocess(ref_rl, ref_data, case["raw"], cfg)
        prepared[name] = data
        arrays = {
            "advantage": np.array([e["advantages"][0] for e in data], dtype=np.float64),
            "group_tokens": np.array([e["group_tokens"][0] for e in data], dtype=np.float64),
            "overflow": np.array([e["overflow"][0] for e in data], dtype=np.float64),
            "num_labels": np.array([e["num_labels"][0] for e in data], dtype=np.float64),
        }
        # per-token lists are constant broadcasts of the scalars above: check, do not store
        for e in data:
            for col in ("advantages", "group_tokens", "overflow", "num_labels"):
                assert len(e[col]) == len(e["input_ids"]) and all(x == e[col][0] for x in e[col])
        n = len(data)
        plans = {
            "all_sp1": (list(range(n)), 1),
            "first5_sp4": (list(range(min(5, n))), 4),
            "tail3_sp1": (list(range(n - 3, n)), 1),
        }
        for pname, (idxs, sp) in plans.items():
            b = ref_data.collate_packed([data[i] for i in idxs], Tok(EOS), seq_parallel=sp)
            for k, v in batch_to_np(b).items():
                arrays[f"packed/{pname}/{k}"] = v
            arrays[f"packed/{pname}/__idx"] = np.asarray(idxs)
            arrays[f"packed/{pname}/__seq_parallel"] = np.asarray(sp)
        for side in ("right", "left"):
            idxs = list(range(min(4, n)))
            # reference collate() needs identical keys in every example; finish_reason is optional
            exs = [{k: v for k, v in data[i].items() if k != "finish_reason"} for i in idxs]
            b = ref_data.collate(exs, Tok(EOS, side))
            for k, v in batch_to_np(b).items():
                arrays[f"padded/{side}/{k}"] = v
            arrays[f"padded/{side}/__idx"] = np.asarray(idxs)
        np.savez_compressed(HERE / f"preprocess_{name}.npz", **arrays)
        (HERE / f"preprocess_{name}.json").write_text(
            json.dumps({"raw": case["raw"], "divide_advantage_by_std": case["divide_advantage_by_std"], "eos_token_id": EOS})
        )
        print(f"preprocess_{name}: {n} sequences")
    return prepared


class FakeModel(torch.nn.Module):
    def __init__(self, logits):
        super().__init__()
        self.logits = torch.nn.Parameter(logits)

    def forward(self, **kwargs):
        return types.SimpleNamespace(logits=self.logits)


class FakeValueModel(FakeModel):
    """What rl_step sees of AutoModelForCausalLMWithValueHead (finetune/value_model.py:54-116): a `value_head`
    attribute and `outputs.value` of shape [B, L]."""

    def __init__(self, logits, value):
        super().__init__(logits)
        self.value_head = torch.nn.Identity()
        self.value = torch.nn.Parameter(value)

    def forward(self, **kwargs):
        return types.SimpleNamespace(logits=self.logits, value=self.value)


RL_CASES = {
    # name: (builder kwargs, RLConfig kwargs, (current_step, max_step))
    "c0_ppo": (dict(vocab=97), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, clamp_log_ratio_ref_new_value=5, batch_size=16, divide_advantage_by_std=False), (0, 10)),
    "c1_ppo_kl_temp": (dict(vocab=128, with_ref=True), dict(policy_loss="ppo", epsilon_low=0.1, epsilon_high=0.3, kl_coef=0.3, final_kl_coef=0.1, clamp_log_ratio_ref_new_value=0.05, temperature=0.7, batch_size=8), (3, 10)),
    "c2_reinforce": (dict(vocab=97, with_ref=True), dict(policy_loss="reinforce", epsilon_high=0.02, kl_coef=0.1, final_kl_coef=0.1, clamp_log_ratio_ref_new_value=5, batch_size=16), (1, 4)),
    "c3_entropy": (dict(vocab=64), dict(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, entropy_bonus=0.01, final_entropy_bonus=0.02, batch_size=16), (2, 8)),
    "c4_groupnorm_overlong": (dict(vocab=97, with_ref=True), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.001, final_kl_coef=0.001, group_normalization=True, overlong_filtering=True, batch_size=16), (0, 10)),
    "c5_rewards_relu": (dict(vocab=97), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, use_advantages=False, relu_log_p_weights=True, batch_size=4), (0, 10)),
    "c6_unpacked": (dict(vocab=97, with_ref=True, unpacked=True), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.001, final_kl_coef=0.001, batch_size=16), (0, 10)),
    "c7_sentinel": (dict(vocab=64, sentinel=True), dict(policy_loss="ppo", kl_coef=0.0, final_kl_coef=0.0, batch_size=16), (0, 10)),
    "c8_offpolicy": (dict(vocab=97, with_ref=True, on_policy=False), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05, clamp_log_ratio_ref_new_value=5, batch_size=16), (0, 10)),
    "c9_sp_padding": (dict(vocab=97, seq_parallel=8), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, batch_size=16), (0, 10)),
    "c10_reinforce_rewards": (dict(vocab=64), dict(policy_loss="reinforce", epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, use_advantages=False, batch_size=16, temperature=1.3), (0, 10)),
    "c11_gspo": (dict(vocab=97, with_ref=True, seed_offset=40), dict(policy_loss="gspo", epsilon_low=0.05, epsilon_high=0.05, kl_coef=0.1, final_kl_coef=0.1, batch_size=16), (0, 10)),
    "c12_gspo_groupnorm_sp": (dict(vocab=64, seq_parallel=8, on_policy=True), dict(policy_loss="gspo", epsilon_low=0.002, epsilon_high=0.002, kl_coef=0.0, final_kl_coef=0.0, group_normalization=True, overlong_filtering=True, batch_size=16), (0, 10)),
    "c13_ppo_all_terms": (dict(vocab=128, with_ref=True), dict(policy_loss="ppo", epsilon_low=0.03, epsilon_high=0.05, kl_coef=0.2, final_kl_coef=0.05, entropy_bonus=0.02, final_entropy_bonus=0.0, temperature=0.9, clamp_log_ratio_ref_new_value=0.08, batch_size=12), (4, 9)),
    "c14_reinforce_all_terms": (dict(vocab=64, with_ref=True, seed_offset=7), dict(policy_loss="reinforce", epsilon_high=0.01, kl_coef=0.05, final_kl_coef=0.05, entropy_bonus=0.005, final_entropy_bonus=0.005, group_normalization=True, temperature=1.2, batch_size=6), (1, 3)),
    "c15_unpacked_left_reinforce": (dict(vocab=97, with_ref=True, unpacked=True, padding_side="left"), dict(policy_loss="reinforce", epsilon_high=0.2, kl_coef=0.01, final_kl_coef=0.01, batch_size=16), (0, 10)),
    "c16_ppo_final_step": (dict(vocab=97, with_ref=True, seed_offset=3), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.28, kl_coef=0.5, final_kl_coef=0.0, entropy_bonus=0.0, final_entropy_bonus=0.03, batch_size=16), (10, 10)),
    "c17_ppo_overlong_only": (dict(vocab=64, seed_offset=11), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, overlong_filtering=True, divide_advantage_by_std=False, batch_size=16), (0, 10)),
    # value-head (actor-critic) branch, rl/__init__.py:265-272, 367-381, 441-448: advantages := rewards - V, + value loss, 5 more stats
    "c18_ppo_value_head": (dict(vocab=97, with_ref=True, seed_offset=21, value_head=True), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.01, final_kl_coef=0.01, value_loss_coef=0.1, batch_size=16), (0, 10)),
    "c19_reinforce_value_head_unpacked": (dict(vocab=64, with_ref=True, unpacked=True, padding_side="left", seed_offset=22, value_head=True), dict(policy_loss="reinforce", epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, overlong_filtering=True, value_loss_coef=0.5, entropy_bonus=0.01, final_entropy_bonus=0.01, batch_size=16), (2, 10)),
    "c20_ppo_value_head_rewards_sp": (dict(vocab=64, seq_parallel=8, seed_offset=23, value_head=True), dict(policy_loss="ppo", epsilon_low=0.1, epsilon_high=0.1, kl_coef=0.0, final_kl_coef=0.0, use_advantages=False, relu_log_p_weights=True, value_loss_coef=1.0, temperature=0.8, batch_size=8), (0, 10)),
    "c21_gspo_value_head": (dict(vocab=97, with_ref=True, seed_offset=24, value_head=True), dict(policy_loss="gspo", epsilon_low=0.05, epsilon_high=0.05, kl_coef=0.05, final_kl_coef=0.05, value_loss_coef=0.2, batch_size=16), (0, 10)),
    "c22_ppo_value_head_groupnorm_overlong": (dict(vocab=64, with_ref=True, seed_offset=25, value_head=True), dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.02, final_kl_coef=0.0, group_normalization=True, overlong_filtering=True, value_loss_coef=0.3, entropy_bonus=0.01, final_entropy_bonus=0.0, batch_size=16), (5, 10)),
    "c23_sentinel_value_head": (dict(vocab=64, sentinel=True, value_head=True), dict(policy_loss="ppo", kl_coef=0.0, final_kl_coef=0.0, value_loss_coef=0.5, batch_size=16), (0, 10)),
}


def gen_rl_step(ref_rl, ref_data, ref_utils, only=None, index=None):
    from pipelinerl_amd.synthetic import make_entries

    for i, (name, (bk, ck, (cur, mx))) in enumerate(RL_CASES.items()):
        if only is not None and name not in only:
            continue
        torch.manual_seed(100 + i)
        V = bk["vocab"]
        cfg = ref_rl.RLConfig(**ck)
        if bk.get("sentinel"):
            batch = ref_utils.create_sentinel_batch(device=None, tokenizer=Tok(EOS), model_version=3)
        else:
            raw = make_entries(2, attempts=3, seq_length=40, vocab=V, seed=200 + i + bk.get("seed_offset", 0), prompt_min=3, prompt_max=9,
                               with_ref=bk.get("with_ref", False))
            data = ref_preprocess(ref_rl, ref_data, raw, cfg)
            if bk.get("unpacked"):
                batch = ref_data.collate([{k: v for k, v in e.items() if k != "finish_reason"} for e in data[:4]], Tok(EOS, bk.get("padding_side", "right")))
            else:
                batch = ref_data.collate_packed(data, Tok(EOS), seq_parallel=bk.get("seq_parallel", 1))
        B, L = batch.input_ids.shape
        logits = (torch.randn(B, L, V) * 2.0).float()
        if bk.get("on_policy", True) and not bk.get("sentinel"):
            # make the behaviour policy close to the current one: old = new + N(0, 0.02) on targets
            z = logits[:, :-1] / cfg.temperature
            nlp = torch.log_softmax(z, -1).gather(2, batch.input_ids[:, 1:, None])[..., 0]
            noise = torch.randn_like(nlp) * 0.02
            old = batch.old_logprobs.clone()
            old[:, 1:] = torch.where(batch.labels[:, 1:] != -100, nlp + noise, old[:, 1:])
            batch.old_logprobs = old
            if bk.get("with_ref"):
                ref = batch.ref_logprobs.clone()
                ref[:, 1:] = torch.where(batch.labels[:, 1:] != -100, nlp + torch.randn_like(nlp) * 0.05, ref[:, 1:])
                batch.ref_logprobs = ref
            else:
                batch.ref_logprobs = old.clone()
        value = None
        if bk.get("value_head"):
            # V(s) near the rewards (binary here) with noise, as a partly trained critic would give; fp32 like the rewards
            value = (0.4 + 0.3 * torch.randn(B, L)).float()
            model = FakeValueModel(logits.clone(), value.clone())
        else:
            model = FakeModel(logits.clone())
        loss, stats = ref_rl.rl_step(model, batch, cur, mx, cfg)
        loss.backward()
        arrays = {f"batch/{k}": v for k, v in batch_to_np(batch).items()}
        if value is not None:
            arrays["value"] = value.numpy()
            arrays["grad_value"] = model.value.grad.numpy()
        arrays["logits"] = logits.numpy()
        arrays["loss"] = np.asarray(loss.item(), dtype=np.float64)
        arrays["grad_logits"] = model.logits.grad.numpy()
        arrays["stats_keys"] = np.asarray(list(stats.keys()))
        arrays["stats_values"] = np.asarray([float(v) for v in stats.values()], dtype=np.float64)
        arrays["config_json"] = np.asarray(json.dumps(ck))
        arrays["steps"] = np.asarray([cur, mx])
        np.savez_compressed(HERE / f"rl_step_{name}.npz", **arrays)
        print(f"rl_step_{name}: shape {tuple(batch.input_ids.shape)} V={V} loss={loss.item():.6g} n_stats={len(stats)}")


def gen_sentinel(ref_utils):
    b = ref_utils.create_sentinel_batch(device=None, tokenizer=Tok(7), model_version=5)
    np.savez_compressed(HERE / "sentinel.npz", **batch_to_np(b))
    ex = ref_utils.create_sentinel_example(3, tokenizer=Tok(7), model_version=9)
    (HERE / "sentinel_example.json").write_text(json.dumps(ex))


def gen_world():
    """GPU partition arithmetic: call the reference's WorldMap._split_gpus_by_purpose (world.py:143-192)
    on a stub `self` (the constructor needs hydra configs and placement state we do not model)."""
    import pipelinerl.world as ref_world

    out = []
    for total in (1, 2, 4, 8):
        for af, pf, ff in ((4, 0, 4), (2, 0, 6), (6, 0, 2), (4, 2, 2), (1, 0, 1), (0.5, 0.25, 3)):
            for tp, pp in ((1, 1), (2, 1), (2, 2)):
                for replicas in (1, 2):
                    me = types.SimpleNamespace(world_size=1, node_size=total, gpus_per_llm=tp * pp, _log_info=lambda x: None)
                    cfg = types.SimpleNamespace(world=types.SimpleNamespace(actor_fraction=af, preprocessor_fraction=pf, finetune_fraction=ff, replicas=replicas))
                    rec = dict(total=total, actor_fraction=af, preprocessor_fraction=pf, finetune_fraction=ff, tp=tp, pp=pp, replicas=replicas)
                    try:
                        ref_world.WorldMap._split_gpus_by_purpose(me, cfg)
                        rec.update(error=None, total_finetune_gpus=me.total_finetune_gpus, gpus_per_actor=me.gpus_per_actor,
                                   gpus_per_preprocessor=me.gpus_per_preprocessor, llms_per_actor=me.llms_per_actor,
                                   total_actor_llms=me.total_actor_llms, weight_update_group_size=me.weight_update_group_size)
                    except ValueError as e:
                        rec.update(error=str(e))
                    out.append(rec)
    (HERE / "world.json").write_text(json.dumps(out))
    print(f"world: {len(out)} partitions, {sum(r['error'] is not None for r in out)} rejected")


def gen_text_path(ref_data):
    """SFT/text branch helpers of preprocess_fn: mask_labels / validate_spans (data.py:47-108) and
    preprocess_fn on a text entry with a toy whitespace tokenizer."""
    rng = np.random.default_rng(3)

    class ToyTok:
        eos_token_id, padding_side = 2, "right"

        def __call__(self, text, return_offsets_mapping=True, max_length=None, truncation=True):
            ids, offs, pos = [], [], 0
            for w in text.split(" "):
                if w:
                    ids.append(3 + (sum(map(ord, w)) % 50))
                    offs.append((pos, pos + len(w)))
                pos += len(w) + 1
            if max_length is not None:
                ids, offs = ids[:max_length], offs[:max_length]
            return {"input_ids": ids, "attention_mask": [1] * len(ids), "offset_mapping": offs}

    cases = []
    for c in range(8):
        words = ["w" * int(rng.integers(1, 6)) for _ in range(int(rng.integers(3, 12)))]
        text = " ".join(words)
        enc = ToyTok()(text)
        n_spans = int(rng.integers(1, 3))
        cuts = sorted(rng.choice(np.arange(len(text) + 1), size=2 * n_spans, replace=False).tolist())
        spans = [(cuts[2 * i], cuts[2 * i + 1]) for i in range(n_spans)]
        labels, mids = ref_data.mask_labels(enc["input_ids"], enc["offset_mapping"], spans)
        entry = {"text": text, "predicted_spans": spans} if c % 2 else {"text": text, "n_predicted": int(rng.integers(0, len(text) + 1))}
        sl = int(rng.integers(3, 20))
        out = ref_data.preprocess_fn(dict(entry), ToyTok(), seq_length=sl, is_rl=False)
        cases.append({"text": text, "spans": spans, "input_ids": enc["input_ids"], "offset_mapping": enc["offset_mapping"],
                      "labels": labels, "midpoints": mids, "entry": entry, "seq_length": sl,
                      "preprocess": {k: out[k] for k in ("input_ids", "labels", "attention_mask")}})
    bad = []
    for spans in ([(-1, 2)], [(0, 99)], [(3, 1)], [(0, 4), (2, 6)]):
        try:
            ref_data.validate_spans("hello world", spans)
            bad.append({"spans": spans, "error": None})
        except ValueError as e:
            bad.append({"spans": spans, "error": str(e)[:20]})
    (HERE / "text_path.json").write_text(json.dumps({"cases": cases, "invalid": bad}))
    print("text_path:", len(cases), "cases")


def main():
    ref_rl, ref_data, ref_utils = import_reference()
    if "--text-only" in sys.argv:
        gen_text_path(ref_data)
        return
    if "--only-new" in sys.argv:  # add cases without rewriting the committed fixtures
        global RL_CASES
        have = {p.stem[len("rl_step_"):] for p in HERE.glob("rl_step_*.npz")}
        idx = {name: i for i, name in enumerate(RL_CASES)}
        todo = {k: v for k, v in RL_CASES.items() if k not in have}
        gen_rl_step(ref_rl, ref_data, ref_utils, only=todo, index=idx)
        return
    gen_preprocess(ref_rl, ref_data)
    gen_rl_step(ref_rl, ref_data, ref_utils)
    gen_sentinel(ref_utils)
    gen_world()
    gen_text_path(ref_data)


if __name__ == "__main__":
    main()
