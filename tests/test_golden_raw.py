"""The raw dumps under tests/golden/raw/ (what tests/golden/replay.jl feeds to the Julia reference) must be the same
bytes as the .npz fixtures the GPU tests replay, laid out as the manifest says, in Julia's column-major order."""
import glob
import os
import re

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NP = {"UInt8": np.uint8, "Int32": np.int32, "Int64": np.int64, "Float64": np.float64}


def _manifest(path):
    """A reader for the tiny TOML subset export_raw.py writes (python 3.10 has no tomllib)."""
    top, arrays, cur = {}, {}, None
    for line in open(path).read().splitlines():
        line = line.strip()
        if not line:
            continue
        m = re.match(r"\[arrays\.(\w+)\]", line)
        if m:
            cur = arrays.setdefault(m.group(1), {})
            continue
        k, v = [x.strip() for x in line.split("=", 1)]
        val = v.strip('"') if v.startswith('"') else ([int(x) for x in v.strip("[]").split(",")] if v.startswith("[") else float(v) if ("." in v or "e" in v) else int(v))
        (cur if cur is not None else top)[k] = val
    return top, arrays


def test_raw_dumps_equal_the_npz_fixtures():
    cases = sorted(glob.glob(os.path.join(HERE, "*.npz")))
    assert len(cases) >= 8
    for f in cases:
        case = os.path.splitext(os.path.basename(f))[0]
        z = np.load(f)
        top, arrays = _manifest(os.path.join(HERE, "raw", case, "manifest.toml"))
        assert top["case"] == case and top["s"] == z["shape"][0] and top["n"] == z["shape"][1]
        assert float(top["per"]) == float(z["per"]) and top["max_iters"] == int(z["max_iters"]) and top["batch"] == z["syndromes"].shape[0]
        for name, spec in arrays.items():
            a = np.fromfile(os.path.join(HERE, "raw", case, spec["file"]), dtype=np.dtype(NP[spec["eltype"]]).newbyteorder("<"))
            want = np.ascontiguousarray(z[name])
            assert list(reversed(want.shape)) == spec["dims"]            # Julia order = reversed C order
            assert a.tobytes() == want.astype(want.dtype.newbyteorder("<")).tobytes(), (case, name)
        assert set(arrays) == {"colptr", "rowval", "syndromes", "errors", "converged", "iters", "llr"}


def test_replay_script_is_shipped_and_flagged_unexecuted():
    src = open(os.path.join(HERE, "replay.jl")).read()
    assert "NOT EXECUTED" in src and "LDPCDecoders.decode!" in src and "manifest.toml" in src
