#!/usr/bin/env python3
"""Copies the reference's packaged DATA files (draws Parquet + meta JSON; no source) into tests/golden/corpus.

    python tests/golden/make_corpus_fixture.py [/root/reference]

57 `*.draws.parquet` (6 of the 63 published models are absent from the checkout, `.MISSING_LARGE_BLOBS`) and all 63
`*.meta.json`, whose `diagnostics` / `checks` blocks are the reference's own goldens for the hot path
(written by convert.py:44-59).  The files are copied byte for byte; the sha256 of each is listed in
tests/golden/corpus/SHA256SUMS, and the reference's own `provenance_manifest.json` (sha256 per packaged file,
generate.py:148-155) is copied beside them: tests/test_oracle_full_corpus.py holds the copies to it.
"""
from __future__ import annotations

import hashlib
import shutil
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent


def main() -> None:
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    data = ref / "packages" / "mcmc-ref-data" / "src" / "mcmc_ref_data" / "data"
    out = HERE / "corpus"
    sums = []
    for sub, pat in (("draws", "*.draws.parquet"), ("meta", "*.meta.json")):
        (out / sub).mkdir(parents=True, exist_ok=True)
        for f in sorted((data / sub).glob(pat)):
            shutil.copyfile(f, out / sub / f.name)
            sums.append(f"{hashlib.sha256(f.read_bytes()).hexdigest()}  {sub}/{f.name}")
    shutil.copyfile(data / "provenance_manifest.json", out / "provenance_manifest.json")   # the reference's own sha256 list
    (out / "SHA256SUMS").write_text("\n".join(sums) + "\n")
    print(f"{len(sums)} files -> {out}")


if __name__ == "__main__":
    main()
