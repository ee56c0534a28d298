"""tools/ are developer scripts (timing, tuning, one-off stress checks), not product and not parity tests --
but they must not rot: every one compiles, answers --help without touching the GPU, and only names
things that exist (attributes of the `quantization_amd` package, entry points the C header declares,
`qamd_dev_*` hooks the developer build defines)."""
import ast
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = sorted(f for f in os.listdir(os.path.join(ROOT, "tools")) if f.endswith(".py"))


def test_every_tool_is_listed_in_the_readme():
    readme = open(os.path.join(ROOT, "tools", "README.md")).read()
    missing = [t for t in TOOLS if t not in readme]
    assert not missing, missing


@pytest.mark.parametrize("tool", TOOLS)
def test_tool_compiles_answers_help_and_names_only_what_exists(tool):
    import quantization_amd as qa
    from quantization_amd import _lib

    path = os.path.join(ROOT, "tools", tool)
    src = open(path).read()
    tree = ast.parse(src)
    assert ast.get_docstring(tree), "a tool says what it is for"
    res = subprocess.run([sys.executable, path, "--help"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert res.returncode == 0 and res.stdout.strip(), res.stderr[-500:]
    # package attributes:  qa.<name>
    for name in set(re.findall(r"\bqa\.([A-Za-z_]\w*)", src)):
        assert hasattr(qa, name), f"{tool}: quantization_amd has no attribute {name}"
    # C entry points: product symbols come from the header, qamd_dev_* from the -DQAMD_DEV sources
    declared = set(_lib.declared_symbols())
    csrc = os.path.join(ROOT, "quantization_amd", "csrc")
    dev_src = "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))
    for sym in set(re.findall(r"\b(qamd_\w+)\b", src)):
        if sym.startswith("qamd_dev_"):
            assert re.search(r"\b%s\s*\(" % sym, dev_src), f"{tool}: no developer hook {sym} in csrc/"
        elif sym.startswith("qamd_probe_"):
            assert sym in open(os.path.join(ROOT, "tools", "probe", "stream_read.hip")).read()
        else:
            assert sym in declared, f"{tool}: {sym} is not declared in include/quantization_amd.h"
    # methods called on the mirror's objects: every .method( that looks like an API call must exist on some class
    api = set()
    for cls in (qa.EncodedVectorsU8, qa.EncodedVectorsBin, qa.EncodedVectorsPQ, qa.ShardedVectorsU8,
                qa.ShardedVectorsBin, qa.ShardedVectorsPQ):
        api.update(dir(cls))
    api.update(dir(qa))  # module-level functions such as qa.topk_scores
    if re.search(r"^\s*from oracle import qoracle", src, re.M):  # a tool that times the oracle's CPU loop beside the GPU's
        from oracle import qoracle
        api.update(dir(qoracle))
    for name in set(re.findall(r"\.(encode\w*|score_\w+|topk\w*|storage_\w+|from_storage|shard\w*)\(", src)):
        assert name in api or name in ("encode",), f"{tool}: no API method {name}"


@pytest.mark.gpu
def test_ann_protocol_harness_small(tmp_path):
    """tools/ann_protocol.py at a small size: the reference's per-query latency statistics and same_10/20/30 accuracy
    (demos/src/ann_benchmark_data.rs:93-185,202-220) for every quantizer, with the oracle's CPU loop beside them."""
    import json
    out = tmp_path / "ann.jsonl"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ann_protocol.py"), "--rows", "30000", "--dims", "64",
                          "--queries", "25", "--cpu-queries", "2", "--out", str(out)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    recs = [json.loads(ln) for ln in open(out)]
    assert len(recs) == 2 * 5  # two metrics x (u8, u8 quantile, pq strided, pq random, binary)
    for r in recs:
        g, c = r["gpu"], r["cpu_oracle_loop"]
        assert g["queries"] == 25 and g["min_ms"] <= g["avg_ms"] <= g["max_ms"] and g["min_ms"] <= g["p95_ms"] <= g["max_ms"]
        assert 0 <= g["same_10"] <= g["same_20"] <= g["same_30"] <= 10
        assert c["topk_scores_equal_the_gpus"] is True and c["queries"] == 2
        if r["quantizer"].startswith("u8"):
            assert g["same_30"] >= 9.0, r  # scalar quantization keeps the ten true neighbours within the first 30
    strided = [r for r in recs if "strided" in r["quantizer"]]
    rand = [r for r in recs if "RANDOM" in r["quantizer"]]
    for a, b in zip(strided, rand):  # the strided k-means sample is no worse than a random one (within noise)
        assert a["gpu"]["same_30"] >= b["gpu"]["same_30"] - 1.0
