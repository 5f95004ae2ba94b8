"""No shipped hot kernel may spill: scratch (private-segment) traffic on a streaming kernel is the difference between 0.64 and 0.35
of the HBM peak (measured in round 5 on the pack kernel: a 32-bit field type in a per-lane struct defeated the compiler's scalar
replacement, the struct went to scratch, 553 -> 998 us; `profiles/r05m_*`).  The translation units are compiled to gfx950 assembly
(device only, no GPU needed) and every kernel's `.amdhsa_private_segment_fixed_size` is checked, with the register budgets the launch
geometries rely on."""

import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

# instantiations that may use scratch: not a default of any dispatch (reachable through the diagnostic tuning table only)
ALLOWED_SCRATCH = ("fused_logits_loss_keep_kernelINS_4BF16E",)  # the row-resident shape on bf16 rows: the bf16 default is the two-sweep kernel


def _checker():
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", ROOT / "scripts" / "check_mfma_hazards.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def resources():
    chk = _checker()
    text = chk.compile_to_asm(("prl_logprob.hip", "prl_loss.hip", "prl_pack.hip", "prl_copy.hip", "prl_value.hip"))
    return chk.kernel_resources(text)


def test_no_hot_kernel_uses_scratch(resources):
    assert len(resources) >= 30, sorted(resources)
    spilled = {k: v for k, v in resources.items() if v["scratch"] and not any(a in k for a in ALLOWED_SCRATCH)}
    assert not spilled, spilled


def test_register_budgets_of_the_launch_geometries(resources):
    def one(*parts):
        got = [v for k, v in resources.items() if all(p in k for p in parts)]
        assert len(got) == 1, (parts, [k for k in resources if parts[0] in k])
        return got[0]

    # the row-resident fused kernel: 1024 threads = 4 waves per SIMD -> 128 registers each, and it needs all of them for 16 resident vectors
    fused = one("fused_logits_loss_keep_kernelINS_3F32ELi1024ELi2ELi16ELi9")
    assert fused["vgpr"] <= 128 and fused["scratch"] == 0
    # the pack kernel that ships (2 tokens per lane, non-temporal stores): 5 waves per SIMD
    pack = one("pack_collate_kernelILb1ELb1ELi2E")
    assert pack["vgpr"] <= 96
    # the step-level loss launch (4 tokens per lane with the gradient store) fits two waves per SIMD
    loss = one("grpo_loss_partial_kernelILi4ELb1ELb1E")
    assert loss["vgpr"] <= 256
    assert one("segment_copy_kernel")["vgpr"] <= 64


def test_the_fused_head_kernels_do_not_spill_either():
    """Every kernel of the fused head (the two translation units with the hand-placed MFMA streams): zero scratch, and the cores that
    run two waves per SIMD stay inside their 256 registers.  Before the DMA sources became `SGPR base + 32-bit lane offset`
    (prl_lmhead_core.h `dma_src`) the dual-plane forward and d-logits kernels kept 16 address pairs in scratch and reloaded three of
    them inside the contraction loop (forward 13.14 -> 12.75 ms once they were gone, profiles/r05t_*)."""
    chk = _checker()
    res = chk.kernel_resources(chk.compile_to_asm())
    assert len(res) >= 15, sorted(res)
    assert not {k: v for k, v in res.items() if v["scratch"]}, {k: v for k, v in res.items() if v["scratch"]}
    assert all(v["vgpr"] <= 256 for v in res.values())
    dual = [v for k, v in res.items() if "CfgDual" in k]
    assert len(dual) == 2 and all(v["vgpr"] <= 248 for v in dual), dual  # a few registers of slack: the next fragment set must not tip them over
