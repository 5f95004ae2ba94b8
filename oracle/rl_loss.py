"""Oracle for the learner loss: numpy fp32 restatement of reference
pipelinerl/finetune/rl/__init__.py:136-450 (`rl_step` after the model call) and
pipelinerl/finetune/rl/utils.py:26-92 (`mask_sum`, `sum_sum`), with the backward pass in closed
form instead of autograd.

All arrays live on the reference's SHIFTED axis: index t in [0, L-1) predicts token t+1.
"""

from __future__ import annotations

from typing import Any

import numpy as np

F32 = np.float32


# RLConfig defaults (rl/__init__.py:43-105): plain dict configs only carry the overrides
DEFAULTS = {
    "policy_loss": "ppo", "use_advantages": True, "epsilon_low": 0.2, "epsilon_high": 0.2, "batch_size": 0,
    "kl_coef": 0.1, "final_kl_coef": 0.1, "entropy_bonus": 0.0, "final_entropy_bonus": 0.0,
    "relu_log_p_weights": False, "clamp_log_ratio_ref_new_value": 10, "divide_advantage_by_std": True,
    "overlong_filtering": False, "group_normalization": False, "temperature": 1.0,
}


def _cfg(config: Any, name: str, default: Any = None) -> Any:
    fallback = DEFAULTS.get(name, default)
    if isinstance(config, dict):
        return config.get(name, fallback)
    return getattr(config, name, fallback)


def linear_decay(current_step: int, max_step: int, initial: float, final: float) -> float:
    """rl/__init__.py:119-133."""
    return initial + (final - initial) * current_step / max_step


def logprob_entropy(logits: np.ndarray, input_ids: np.ndarray, temperature: float):
    """rl/__init__.py:207-233.  logits [B, L, V] fp32, input_ids [B, L] -> ([B, L-1], [B, L-1], softmax)."""
    z = (logits[:, :-1, :].astype(F32) / F32(temperature)).astype(F32)
    m = z.max(axis=-1, keepdims=True)
    ez = np.exp(z - m, dtype=F32)
    lse = (m[..., 0] + np.log(ez.sum(axis=-1, dtype=F32), dtype=F32)).astype(F32)
    nxt = input_ids[:, 1:]
    sel = np.take_along_axis(z, nxt[..., None], axis=-1)[..., 0]
    new_logprobs = (sel - lse).astype(F32)
    logp = (z - lse[..., None]).astype(F32)
    p = np.exp(logp, dtype=F32)
    with np.errstate(invalid="ignore"):
        entropy = (-(p * logp).sum(axis=-1, dtype=F32)).astype(F32)
    return new_logprobs, entropy, p, logp


def segments_from_positions(position_ids: np.ndarray) -> list[tuple[int, int]]:
    """rl/__init__.py:165-185: sequence starts are position_ids == 0 (index 0 forced)."""
    pos = position_ids.reshape(-1)
    starts = np.flatnonzero(pos == 0).tolist()
    if not starts or starts[0] != 0:
        starts = [0] + starts
    bounds = starts + [len(pos)]
    return list(zip(bounds[:-1], bounds[1:]))


def _mask_sum(values: np.ndarray, mask: np.ndarray) -> np.float32:
    """rl/utils.py:26-31: (values * mask).nan_to_num(0).sum() in fp32."""
    with np.errstate(invalid="ignore", over="ignore"):
        v = (values.astype(F32) * mask.astype(F32)).astype(F32)
    return np.nan_to_num(v, nan=0.0).sum(dtype=F32)


def sum_sum(values: np.ndarray, mask: np.ndarray, segments) -> np.float32:
    """rl/utils.py:71-92: per-segment masked sums, then their sum (plain masked sum when unpacked
    or for the 1-wide sentinel case)."""
    if segments and values.shape[-1] != 1:
        parts = np.array([_mask_sum(values[0, s:e], mask[0, s:e]) for s, e in segments], dtype=F32)
        return parts.sum(dtype=F32)
    return _mask_sum(values, mask)


def token_loss(batch: dict[str, np.ndarray], new_logprobs: np.ndarray, entropy: np.ndarray, config: Any,
               current_step: int, max_step: int, is_packed: bool, value: np.ndarray | None = None) -> dict[str, Any]:
    """rl/__init__.py:238-448.  `batch` holds the unshifted [B, L] arrays of PipelineBatchEncoding.
    Returns loss (fp32), stats dict (or the no-label dict), and the closed-form gradients
    d loss / d new_logprobs, d loss / d entropy on the shifted axis.
    `value`: outputs.value [B, L] of a model with a value head (:162, :265-272, :367-381, :441-448): the advantages become
    rewards - value[:, :-1] (detached), 0.5 (V - reward)^2 w joins the loss with `value_loss_coef`, five more statistics;
    `g_value` [B, L] = d loss / d value."""
    labels = batch["labels"]
    mask = (labels != -100)[:, 1:]
    fm = mask.astype(F32)
    sh = lambda k: batch[k][:, 1:].astype(F32)  # noqa: E731
    rewards, ref, old = sh("rewards"), sh("ref_logprobs"), sh("old_logprobs")
    group_tokens, num_labels, overflow, advantages = sh("group_tokens"), sh("num_labels"), sh("overflow"), sh("advantages")
    if is_packed:
        segments = segments_from_positions(batch["position_ids"][0])
        num_sequences = len(segments)
    else:
        segments = None
        num_sequences = labels.shape[0]

    nlp = new_logprobs.astype(F32)
    if value is not None:
        vp = value[:, :-1].astype(F32)  # no target for the last token (:267)
        with np.errstate(all="ignore"):
            advantages = (rewards - vp).astype(F32)  # (:272)
    with np.errstate(all="ignore"):
        if _cfg(config, "group_normalization"):
            w = (np.ones_like(group_tokens) / group_tokens).astype(F32)
        else:
            w = (np.ones_like(group_tokens) / F32(_cfg(config, "batch_size"))).astype(F32)
        if _cfg(config, "overlong_filtering"):
            w = (w * (F32(1) - overflow)).astype(F32)

        lrno = (nlp - old).astype(F32)
        ratio = np.exp(lrno, dtype=F32)
        lrrn = (ref - nlp).astype(F32)
        A = advantages if _cfg(config, "use_advantages", True) else rewards
        if _cfg(config, "relu_log_p_weights"):
            A = np.maximum(A, F32(0))
        C = F32(_cfg(config, "clamp_log_ratio_ref_new_value"))
        clamp_rn = (np.abs(lrrn) > C)
        cl = np.clip(lrrn, -C, C).astype(F32)
        ecl = np.exp(cl, dtype=F32)
        kl = (ecl - cl - F32(1)).astype(F32)
        kl_no = (np.exp(lrno, dtype=F32) - lrno - F32(1)).astype(F32)
        ent_coef = linear_decay(current_step, max_step, _cfg(config, "entropy_bonus", 0.0), _cfg(config, "final_entropy_bonus", 0.0))
        kl_coef = linear_decay(current_step, max_step, _cfg(config, "kl_coef"), _cfg(config, "final_kl_coef"))
        use_entropy = _cfg(config, "entropy_bonus", 0.0) != 0.0 or _cfg(config, "final_entropy_bonus", 0.0) != 0.0

        lo = F32(1 - _cfg(config, "epsilon_low"))
        hi = F32(1 + _cfg(config, "epsilon_high"))
        algo = _cfg(config, "policy_loss")
        if algo == "ppo":
            s1 = (ratio * A).astype(F32)
            cr = np.clip(ratio, lo, hi).astype(F32)
            clamp_no = cr != ratio
            s2 = (cr * A).astype(F32)
            pol = np.minimum(s1, s2)
            inside = (ratio >= lo) & (ratio <= hi)
            d1 = (A * ratio).astype(F32)
            d2 = np.where(inside, d1, F32(0))
            dpol = np.where(s1 < s2, d1, np.where(s2 < s1, d2, F32(0.5) * d1 + F32(0.5) * d2)).astype(F32)
            ratio_stat = ratio
        elif algo == "reinforce":
            clamp_no = ratio > hi
            crr = np.clip(ratio, F32(0), hi).astype(F32)
            pol = (nlp * A * crr).astype(F32)
            dpol = (A * crr).astype(F32)
            ratio_stat = crr
        elif algo == "gspo":
            # sequence-level ratio (rl/__init__.py:310-352, rl/utils.py:106-208): per-segment masked
            # sums on the shifted axis, segment of shifted position t = segment_ids[t + 1]
            if segments is None:
                raise ValueError("GSPO loss requires packed sequences with segments")
            seg = batch["segment_ids"][0, 1:]
            n_seg = int(seg.max()) + 1 if seg.size else 0
            mm = fm[0]
            cnt = np.zeros(n_seg, dtype=F32)
            lrn_sum = np.zeros(n_seg, dtype=F32)
            adv_sum = np.zeros(n_seg, dtype=F32)
            w_sum = np.zeros(n_seg, dtype=F32)
            np.add.at(cnt, seg, mm)
            np.add.at(lrn_sum, seg, lrno[0] * mm)
            np.add.at(adv_sum, seg, advantages[0] * mm)
            np.add.at(w_sum, seg, w[0] * mm)
            den = np.maximum(cnt, F32(1e-6))
            g_ratio = np.exp(lrn_sum / den, dtype=F32)
            g_adv = (adv_sum / den).astype(F32)
            valid = (cnt > 0) & (w_sum > 0)
            s1 = (g_ratio * g_adv).astype(F32)
            cr = np.clip(g_ratio, lo, hi).astype(F32)
            seg_clamped = (cr != g_ratio) & valid
            s2 = (cr * g_adv).astype(F32)
            if batch.get("sentinel") or n_seg == 0:
                gspo_loss = F32(0)
            else:
                gspo_loss = F32(-(np.minimum(s1, s2) * valid.astype(F32) * w_sum).sum(dtype=F32))
            inside = (g_ratio >= lo) & (g_ratio <= hi)
            dmin = np.where(s1 < s2, g_adv, np.where(s2 < s1, np.where(inside, g_adv, F32(0)),
                                                      F32(0.5) * g_adv + F32(0.5) * np.where(inside, g_adv, F32(0)))).astype(F32)
            coef = (-(w_sum * valid.astype(F32)) * dmin * g_ratio / den).astype(F32)
            clamp_no = seg_clamped[seg][None, :]
            ratio_stat = ratio
            pol = dpol = None
        else:
            raise ValueError(f"Unknown algorithm {algo}")

        if algo == "gspo":
            loss = gspo_loss
            tok = np.zeros_like(w)
            g_nlp = (coef[seg] * mm)[None, :].astype(F32) if not batch.get("sentinel") else np.zeros_like(w)
            g_ent = np.zeros_like(w)
        else:
            tok = (pol - F32(kl_coef) * kl).astype(F32)
            if use_entropy:
                tok = (tok + F32(ent_coef) * entropy.astype(F32)).astype(F32)
            tok = (tok * w).astype(F32)
            loss = F32(-sum_sum(tok, mask, segments))

            # closed-form backward (App. A): nan_to_num passes gradient only where finite
            kl_inside = (lrrn >= -C) & (lrrn <= C)
            dkl = np.where(kl_inside, F32(1) - ecl, F32(0)).astype(F32)
            finite = np.isfinite(tok * fm)
            g_nlp = np.where(finite, -((dpol - F32(kl_coef) * dkl) * w) * fm, F32(0)).astype(F32)
            g_ent = (np.where(finite, -(F32(ent_coef) * w) * fm, F32(0)) if use_entropy else np.zeros_like(tok)).astype(F32)

        if value is not None:
            # (:367-381) value labels are the shifted rewards; sum_sum's nan_to_num passes gradient only where finite
            coef = F32(_cfg(config, "value_loss_coef", 0.0))
            diff = (vp - rewards).astype(F32)
            vl_tok = ((F32(0.5) * (diff * diff).astype(F32)).astype(F32) * w).astype(F32)
            value_loss = F32(sum_sum(vl_tok, mask, segments))
            loss = F32(loss + F32(coef * value_loss))
            g_value = np.zeros(value.shape, dtype=F32)
            g_value[:, :-1] = np.where(np.isfinite(vl_tok * fm), (coef * (diff * w).astype(F32)).astype(F32) * fm, F32(0))

    out: dict[str, Any] = {"loss": loss, "g_nlp": g_nlp, "g_ent": g_ent, "num_sequences": num_sequences,
                           "finite": bool(np.isfinite(nlp).all() and np.isfinite(lrrn).all() and np.isfinite(kl).all() and np.isfinite(loss))}
    if value is not None:
        out["g_value"] = g_value
    input_size = int(batch["input_ids"].size)
    if int(mask.sum()) == 0:
        out["stats"] = {"input_size": float(input_size)}
        return out

    with np.errstate(all="ignore"):
        per = lambda x: float(sum_sum((x.astype(F32) / num_labels).astype(F32), mask, segments))  # noqa: E731
        stats = {
            "loss": float(loss), "max_loss": float(loss), "min_loss": float(loss),
            "reward": per(rewards), "max_reward": float(rewards[mask].max()), "min_reward": float(rewards[mask].min()),
            "entropy": per(entropy), "old_logprobs": per(old), "new_logprobs": per(nlp), "ref_logprobs": per(ref),
            "advantage": per(advantages), "max_advantage": float(advantages[mask].max()), "min_advantage": float(advantages[mask].min()),
            "kl": per(kl), "kl_new_old": per(kl_no), "mean_abs_log_ratio_new_old": per(np.abs(lrno)),
            "max_kl": float(kl[mask].max()), "min_kl": float(kl[mask].min()),
            "ratio_new_old": per(ratio_stat),
            "ratio_new_old_sum": float(sum_sum(ratio_stat, mask, segments)),
            "ratio_new_old_squared_sum": float(sum_sum((ratio_stat * ratio_stat).astype(F32), mask, segments)),
            "ratio_ref_new": per(np.exp(lrrn, dtype=F32)),
            "ratio_ref_old": per(np.exp((ref - old).astype(F32), dtype=F32)),
            "clamp_log_ratio_ref_new_indicator": per(clamp_rn.astype(F32)),
            "clamp_log_ratio_new_old_indicator": per(clamp_no.astype(F32)),
            "token_weight": per(w), "max_token_weight": float(w[mask].max()), "min_token_weight": float(w[mask].min()),
            "kl_coef": num_sequences * kl_coef, "entropy_bonus_coef": num_sequences * ent_coef,
            "num_output_tokens_sum": int(mask.sum()), "input_size": input_size,
        }
        if value is not None:  # (:441-448)
            stats.update({
                "value_mean": per(vp), "value_max": float(vp[mask].max()), "value_min": float(vp[mask].min()),
                "value_loss": float(value_loss), "value_mse": per((diff * diff).astype(F32)),
            })
    out["stats"] = stats
    return out


def rl_step(logits: np.ndarray, batch: dict[str, np.ndarray], config: Any, current_step: int, max_step: int,
            is_packed: bool, value: np.ndarray | None = None) -> dict[str, Any]:
    """Full post-model path: logits (and the value head's predictions, if the model has one) -> loss, stats,
    d loss / d logits [B, L, V] (and `g_value`)."""
    temperature = _cfg(config, "temperature", 1.0)
    nlp, ent, p, logp = logprob_entropy(logits, batch["input_ids"], temperature)
    res = token_loss(batch, nlp, ent, config, current_step, max_step, is_packed, value=value)
    g, gh = res["g_nlp"], res["g_ent"]
    B, L, V = logits.shape
    onehot = np.zeros((B, L - 1, V), dtype=F32)
    np.put_along_axis(onehot, batch["input_ids"][:, 1:, None], F32(1), axis=-1)
    with np.errstate(invalid="ignore"):
        dz = g[..., None] * (onehot - p) - gh[..., None] * (p * (logp + ent[..., None]))
    grad = np.zeros((B, L, V), dtype=F32)
    grad[:, :-1, :] = (dz / F32(temperature)).astype(F32)
    res.update(new_logprobs=nlp, entropy=ent, grad_logits=grad)
    return res
