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
    for name in set(re.findall(r"\.(encode\w*|score_\w+|topk\w*|storage_\w+|from_storage|shard\w*)\(", src)):
        assert name in api or name in ("encode",), f"{tool}: no API method {name}"
