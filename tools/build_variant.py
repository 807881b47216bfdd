#!/usr/bin/env python3
"""A/B builds of the library: `build_variant.py NAME -DMACRO=V ...` -> mcmc-db_amd/lib/variants/libmcmcref_hip_NAME.so
(select with MCMC_REF_HIP_LIB=<path>).  Development aid: several variants travel to the GPU box in one gpurun call."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "mcmc-db_amd"))
import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = B.OUT.parent / "variants" / f"libmcmcref_hip_{name}.so"
out.parent.mkdir(parents=True, exist_ok=True)
cmd = [B.hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall", "-Wno-unused-value",
       "-Wno-pass-failed", *flags, "-o", str(out), str(B.SRC)]
subprocess.run(cmd, check=True)
print(out)
