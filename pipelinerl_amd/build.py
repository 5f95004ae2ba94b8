"""Build libprl.so (HIP kernels + C ABI) in-tree for gfx950.

`python -m pipelinerl_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles for
gfx950 without a GPU.  The shared object lands in pipelinerl_amd/lib/ (git-ignored; it
travels with the working tree to the GPU box).
"""

from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
OBJ_DIR = LIB_DIR / "obj"
LIB_PATH = LIB_DIR / "libprl.so"
INCLUDE = PKG_DIR.parent / "include"

ARCH = os.environ.get("PRL_OFFLOAD_ARCH", "gfx950")

# -ffp-contract=off: the token math mirrors the reference's fp32 op sequence; FMAs are
# requested explicitly (__builtin_fmaf) where wanted.
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}"]
HIP_FLAGS = [f"--offload-arch={ARCH}", "-x", "hip"]

SOURCES = [
    "prl_api.cpp",
    "prl_ring.cpp",
    "prl_log.cpp",
    "prl_publish.cpp",
    "prl_wsync.cpp",
    "prl_ipc.cpp",
    "prl_loss.hip",
    "prl_value.hip",
    "prl_logprob.hip",
    "prl_pack.hip",
    "prl_copy.hip",
    "prl_lmhead_prepare.hip",
    "prl_lmhead_fwd.hip",
    "prl_lmhead_bwd.hip",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libprl.so cannot be built on this machine")
    return exe


def _digest(paths: list[Path], extra: str) -> str:
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _compile_one(hipcc: str, src: Path, obj: Path, headers: list[Path], verbose: bool) -> None:
    flags = list(COMMON_FLAGS)
    if src.suffix == ".hip":
        flags += HIP_FLAGS
    stamp = obj.with_suffix(".sha")
    dig = _digest([src] + headers, " ".join(flags))
    if obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return
    cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    stamp.write_text(dig)


def build(verbose: bool = False, force: bool = False) -> Path:
    """Compile every source and link pipelinerl_amd/lib/libprl.so. Returns its path."""
    hipcc = _hipcc()
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    if force:
        for p in OBJ_DIR.glob("*"):
            p.unlink()
    headers = sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    srcs = [CSRC / s for s in SOURCES]
    objs = [OBJ_DIR / (s.name + ".o") for s in srcs]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        futs = [ex.submit(_compile_one, hipcc, s, o, headers, verbose) for s, o in zip(srcs, objs)]
        for f in futs:
            f.result()
    newest_obj = max(o.stat().st_mtime for o in objs)
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest_obj:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB_PATH), *map(str, objs), "-ldl", "-lrt", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    path = build(verbose=True, force="--force" in sys.argv)
    print(f"built {path}")
