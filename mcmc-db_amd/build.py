#!/usr/bin/env python3
"""Builds mcmc-db_amd/lib/libmcmcref_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "csrc" / "mcr_api.hip"
DEPS = sorted((HERE / "csrc").glob("*.h*")) + [SRC, HERE.parent / "include" / "mcmcref_hip.h"]
OUT = HERE / "lib" / "libmcmcref_hip.so"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def build(force: bool = False, verbose: bool = False) -> Path:
    OUT.parent.mkdir(exist_ok=True)
    if not force and OUT.exists() and all(OUT.stat().st_mtime >= d.stat().st_mtime for d in DEPS):
        return OUT
    cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-value", "-Wno-pass-failed", "-o", str(OUT), str(SRC)]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
