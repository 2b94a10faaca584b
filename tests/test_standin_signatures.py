"""The stand-in headers of tests/cpp/integration/ref/ restate the reference's interfaces so that the INTEGRATION.md snippets compile in the situation they meet in
a checkout.  When the reference tree is present (this container; never on the GPU box) its own headers are parsed and every member-function signature of the
hot path's two classes -- include/ORBmatcher.h:41-83 and the four statics of include/Optimizer.h:50-56 -- is compared with the stand-in's: name, return type,
parameter types in order (parameter names, defaults, `std::` and white space aside).  VERDICT r3 next #8d."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")


def _split(params):
    out, depth, cur = [], 0, ""
    for ch in params:
        if ch in "<(":
            depth += 1
        elif ch in ">)":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _norm_type(p):
    p = p.split("=")[0].strip()                                  # default value
    p = re.sub(r"\bstd::", "", p)
    m = re.match(r"^(.*?[\s\*&>])([A-Za-z_]\w*)$", p)           # trailing parameter name
    if m and m.group(2) not in ("int", "float", "bool", "long", "unsigned", "Mat", "size_t"):
        p = m.group(1)
    p = re.sub(r"\s+", "", p)
    # `const T&` and `T const&` never occur mixed here; `const int` / `const float` by-value qualifiers do not change a declaration
    p = re.sub(r"^const(int|float|bool|unsignedlong)$", r"\1", p)
    return p


def signatures(text, cls):
    """{(name, (param types...)): return type} of the member functions declared in `class cls { ... }`."""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    m = re.search(r"class\s+%s\b[^{;]*\{" % cls, text)
    assert m, cls
    i, depth = m.end(), 1
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    body = text[m.end():i - 1]
    sigs = {}
    for d in re.finditer(r"([\w:\s\*&<>]+?)\b(\w+)\s*\(([^;{}]*)\)\s*;", body):
        ret, name, params = d.group(1), d.group(2), d.group(3)
        ret = re.sub(r"\b(static|inline|virtual|public:|protected:|private:)\b", "", ret)
        ret = re.sub(r"\s+", "", ret.replace("public:", "").replace("protected:", ""))
        if name == cls:
            ret = ""
        sigs[(name, tuple(_norm_type(p) for p in _split(params)))] = ret
    return sigs


def test_orbmatcher_stand_in_declares_the_reference_interface():
    ref = signatures(open(os.path.join(REF, "include", "ORBmatcher.h")).read(), "ORBmatcher")
    mine = signatures(open(os.path.join(ROOT, "tests", "cpp", "integration", "ref", "ORBmatcher.h")).read(), "ORBmatcher")
    assert len(ref) >= 16, sorted(ref)
    assert ref == mine, "only in the reference: %s\nonly in the stand-in: %s" % (sorted(set(ref.items()) - set(mine.items())), sorted(set(mine.items()) - set(ref.items())))
    # the eleven searches + DescriptorDistance + the constructor are all there
    names = [k[0] for k in ref]
    assert names.count("SearchByProjection") == 4 and names.count("SearchByBoW") == 2 and names.count("Fuse") == 2


def test_optimizer_stand_in_declares_the_four_statics_of_the_path():
    ref = signatures(open(os.path.join(REF, "include", "Optimizer.h")).read(), "Optimizer")
    mine = signatures(open(os.path.join(ROOT, "tests", "cpp", "integration", "ref", "Optimizer.h")).read(), "Optimizer")
    want = {k: v for k, v in ref.items() if k[0] in ("BundleAdjustment", "GlobalBundleAdjustemnt", "LocalBundleAdjustment", "PoseOptimization")}
    assert len(want) == 4, sorted(ref)
    assert want == mine, "only in the reference: %s\nonly in the stand-in: %s" % (sorted(set(want.items()) - set(mine.items())), sorted(set(mine.items()) - set(want.items())))
    # this fork's BundleAdjustment takes the map planes (include/Optimizer.h:50)
    assert any(k[0] == "BundleAdjustment" and any("MapPlane" in t for t in k[1]) for k in mine)
