"""Static check of the hand-placed (asm) MFMA streams in csrc/prl_lmhead_core.h (instantiated by prl_lmhead_fwd.hip / prl_lmhead_bwd.hip).

MFMAs written as `asm volatile` are invisible to hipcc's hazard recogniser, so two software-managed hazards of gfx950 have to be
kept out of the instruction stream by construction (csrc: `mfma_pin_acc`, `mfma_settle`) - and this script verifies the generated
ISA instead of trusting that construction:

  W->R  a VALU instruction (v_mov, v_accvgpr_write, v_add, ...) writes a register that an asm MFMA reads (SrcA / SrcB / SrcC)
        fewer than MIN_GAP instructions later.  (Found on hardware: `v_mov_b64 v[56:57], 0` directly in front of the first MFMA
        into v[56:71] left one accumulator register stale.)
  R<-W  a non-MFMA instruction reads or overwrites the destination of an asm MFMA fewer than MIN_GAP instructions after it without
        an `s_nop` run in between (the result is still in flight).

Compiles the file to assembly with hipcc (gfx950) and scans every kernel.  Exit code 1 and a listing when something is found.

    python scripts/check_mfma_hazards.py [--asm file.s]
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
MIN_GAP = 4  # instructions (each >= 1 wait state; an intervening MFMA counts as 4)


SOURCES = ("prl_lmhead_fwd.hip", "prl_lmhead_bwd.hip")  # the translation units that instantiate the hand-placed streams of prl_lmhead_core.h


def compile_to_asm(sources=SOURCES) -> str:
    tmp = Path(tempfile.mkdtemp())
    text = []
    for name in sources:
        out = tmp / (name + ".s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'include'}",
                               f"-I{ROOT / 'pipelinerl_amd' / 'csrc'}", "-S", "--cuda-device-only", str(ROOT / "pipelinerl_amd" / "csrc" / name),
                               "-o", str(out)], stderr=subprocess.DEVNULL)
        text.append(out.read_text())
    return "\n".join(text)


def kernel_resources(text: str) -> dict[str, dict[str, int]]:
    """Per kernel of an assembly listing: scratch bytes per lane (spills / arrays the compiler could not keep in registers), VGPRs,
    AGPRs and LDS bytes, from its `.amdhsa_kernel` block."""
    import re

    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)

        def field(key, default=0):
            f = re.search(rf"\.amdhsa_{key} (\d+)", body)
            return int(f.group(1)) if f else default

        vgpr, accum = field("next_free_vgpr"), field("accum_offset")
        out[m.group(1)] = {"scratch": field("private_segment_fixed_size"), "vgpr": vgpr, "agpr": max(0, vgpr - accum) if accum else 0,
                           "lds": field("group_segment_fixed_size")}
    return out


def regs(tok: str) -> set[str]:
    tok = tok.strip()
    m = re.match(r"([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return {f"{m.group(1)}{k}" for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {f"{m.group(1)}{m.group(2)}"} if m else set()


def scan(text: str) -> list[str]:
    problems: list[str] = []
    kernel = "?"
    window: list[tuple[str, set[str], set[str], bool]] = []  # (text, writes, reads, is_asm_mfma) of the last instructions
    in_asm = False
    for raw in text.split("\n"):
        line = raw.strip()
        m_fn = re.match(r"^(_Z\w+):", raw)
        if m_fn:
            kernel = m_fn.group(1)
            window = []
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            if line.endswith(":") and not line.startswith(";"):
                window = []  # a label: control flow joins, the linear window is over
            continue
        instr = line.split(";")[0].strip()
        parts = instr.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        is_mfma = op.startswith("v_mfma")
        asm_mfma = is_mfma and in_asm
        if op.startswith("s_nop"):
            n = int(ops[0]) + 1 if ops else 1
            window.extend([("s_nop", set(), set(), False)] * min(n, 2 * MIN_GAP))
            window = window[-4 * MIN_GAP:]
            continue
        writes, reads = set(), set()
        if is_mfma:
            writes = regs(ops[0])
            for o in ops[1:4]:
                reads |= regs(o)
        elif op.startswith(("v_", "ds_read", "ds_load", "scratch_load", "global_load", "buffer_load", "flat_load")):
            if op.startswith(("v_cmp", "v_cmpx")):
                for o in ops:
                    reads |= regs(o)
            else:
                writes = regs(ops[0]) if ops else set()
                for o in ops[1:]:
                    reads |= regs(o)
                if "lds" in op:  # global_load_lds: no register destination
                    reads |= writes
                    writes = set()
        else:
            for o in ops:
                reads |= regs(o)
        valu_write = op.startswith("v_") and not is_mfma and bool(writes)
        # W->R: this asm MFMA reads something a recent VALU wrote
        if asm_mfma:
            gap = 0
            for prev_text, prev_w, _prev_r, prev_asm in reversed(window):
                if gap >= MIN_GAP:
                    break
                if prev_text.startswith("v_") and not prev_text.startswith("v_mfma") and prev_w & reads:
                    problems.append(f"{kernel}: VALU write {sorted(prev_w & reads)[:3]} {gap} instruction(s) before an asm MFMA reads it:  {prev_text}  ->  {instr}")
                    break
                gap += 4 if prev_text.startswith("v_mfma") else 1
        # R<-W: a non-MFMA instruction touches the destination of a recent asm MFMA
        if not is_mfma and (reads or writes):
            gap = 0
            for prev_text, prev_w, _prev_r, prev_asm in reversed(window):
                if gap >= MIN_GAP:
                    break
                if prev_asm and prev_w & (reads | writes):
                    problems.append(f"{kernel}: {instr}  touches {sorted(prev_w & (reads | writes))[:3]} {gap} instruction(s) after the asm MFMA that produces it")
                    break
                gap += 4 if prev_text.startswith("v_mfma") else 1
        window.append((instr, writes, reads, asm_mfma))
        window = window[-4 * MIN_GAP:]
    return problems


def main() -> int:
    text = Path(sys.argv[sys.argv.index("--asm") + 1]).read_text() if "--asm" in sys.argv else compile_to_asm()
    n_asm = text.count(";;#ASMSTART")
    problems = scan(text)
    print(f"{n_asm} asm statements scanned, {len(problems)} potential hazards")
    for p in problems[:40]:
        print("  " + p[:400])
    return 1 if problems else 0


if __name__ == "__main__":
    raise SystemExit(main())
