#!/usr/bin/env python3
"""Dependency-free dumps of the golden fixtures for the Julia replay (tests/golden/replay.jl): every array of every
<case>.npz as a raw little-endian file under tests/golden/raw/<case>/ plus a manifest.toml (Julia reads TOML with its
standard library; .npz would need a third-party package).  numpy's C-order [B][s] IS Julia's column-major s x B, so
the files load straight into the reference's matrix orientation.  Run from the repo root after make_golden.py:
`python tests/golden/export_raw.py`."""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
JL = {"uint8": "UInt8", "int32": "Int32", "int64": "Int64", "float64": "Float64"}


def main():
    for f in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        case = os.path.splitext(os.path.basename(f))[0]
        out = os.path.join(HERE, "raw", case)
        os.makedirs(out, exist_ok=True)
        z = np.load(f)
        lines = [f'case = "{case}"',
                 f's = {int(z["shape"][0])}', f'n = {int(z["shape"][1])}',
                 f'per = {float(z["per"])!r}', f'max_iters = {int(z["max_iters"])}',
                 f'batch = {int(z["syndromes"].shape[0])}', ""]
        for name in ("colptr", "rowval", "syndromes", "errors", "converged", "iters", "llr"):
            a = np.ascontiguousarray(z[name])
            a.astype(a.dtype.newbyteorder("<")).tofile(os.path.join(out, name + ".bin"))
            # dims in JULIA order (fastest first): reverse of numpy's C-order shape
            dims = ", ".join(str(d) for d in reversed(a.shape)) if a.ndim else "1"
            lines += [f"[arrays.{name}]", f'file = "{name}.bin"', f'eltype = "{JL[str(a.dtype)]}"', f"dims = [{dims}]", ""]
        open(os.path.join(out, "manifest.toml"), "w").write("\n".join(lines))
        print(case, "->", out)


if __name__ == "__main__":
    main()
