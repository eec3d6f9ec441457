"""The drop-in Python surface, pinned mechanically: every public class, method, function, parameter name and default of the
reference's `models/models.py`, `models/kv_caching.py` and `inference/vitomr_inference.py` - recorded in tests/golden/surface.json by
oracle/gen_surface.py from the reference's files - must exist in the mirror package with the same positional order and defaults.  The mirror
may ADD parameters, but only behind the reference's and only with defaults (e.g. `noises=None` to inject the masking noise)."""
import ast
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "surface.json")))

# Reference callables that are deliberately NOT mirrored: the subprocess post-processing around external CLIs (olimpic_app / musescore3 /
# ImageMagick), which SURVEY section 2 row 3 marks out of scope.  Everything else must be present.
ALLOW_MISSING = {
    ("inference.vitomr_inference", None, "delinearize"),
    ("inference.vitomr_inference", None, "convert_back_to_img"),
}


def _default_equal(ref_src, value):
    if ref_src == "F.gelu":
        import torch.nn.functional as F
        return value is F.gelu
    return ast.literal_eval(ref_src) == value


def _diff_params(where, ref_params, fn):
    problems = []
    sig = inspect.signature(fn)
    mine = [p for p in sig.parameters.values()]
    ref_named = [p for p in ref_params if p[2] in ("pos", "kw")]
    for i, (name, default, kind) in enumerate(ref_named):
        if i >= len(mine) or mine[i].name != name:
            problems.append(f"{where}: parameter {i} should be `{name}`, mirror has `{mine[i].name if i < len(mine) else None}`")
            continue
        have = mine[i].default
        if default is None:
            if have is not inspect.Parameter.empty:
                problems.append(f"{where}: `{name}` has no default in the reference, mirror gives {have!r}")
        elif have is inspect.Parameter.empty or not _default_equal(default, have):
            problems.append(f"{where}: `{name}` default should be {default}, mirror has {have!r}")
    for p in mine[len(ref_named):]:
        if p.default is inspect.Parameter.empty and p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY):
            problems.append(f"{where}: extra mirror parameter `{p.name}` has no default (callers written for the reference would break)")
    return problems


def _collect():
    problems, checked = [], 0
    for modname, mod in SURFACE.items():
        m = importlib.import_module("acai_omr_amd." + modname)
        for const in mod["constants"]:
            if not hasattr(m, const):
                problems.append(f"{modname}: constant {const} missing")
        for fname, params in mod["functions"].items():
            if (modname, None, fname) in ALLOW_MISSING:
                continue
            fn = getattr(m, fname, None)
            if fn is None:
                problems.append(f"{modname}.{fname}: missing")
                continue
            checked += 1
            problems += _diff_params(f"{modname}.{fname}", params, fn)
        for cname, c in mod["classes"].items():
            cls = getattr(m, cname, None)
            if cls is None:
                problems.append(f"{modname}.{cname}: class missing")
                continue
            for base in c["bases"]:
                bname = base.split(".")[-1]
                if not any(k.__name__ == bname for k in cls.__mro__[1:]):
                    problems.append(f"{modname}.{cname}: should derive from {base}")
            for pname in c["properties"]:
                if not isinstance(inspect.getattr_static(cls, pname, None), property):
                    problems.append(f"{modname}.{cname}.{pname}: should be a property")
            for mname, params in c["methods"].items():
                if (modname, cname, mname) in ALLOW_MISSING:
                    continue
                fn = inspect.getattr_static(cls, mname, None)
                if fn is None:
                    problems.append(f"{modname}.{cname}.{mname}: missing")
                    continue
                checked += 1
                problems += _diff_params(f"{modname}.{cname}.{mname}", params, getattr(cls, mname))
    return problems, checked


def test_mirror_surface_matches_reference():
    problems, checked = _collect()
    assert checked >= 65, checked
    assert not problems, "\n".join(problems)


def test_subclass_overrides_are_the_subclass_contract():
    """A method the reference overrides in a subclass must not silently resolve to the base class's in the mirror (round 2: MAEEncoder.batchify
    returned Encoder.batchify's 2-tuple).  For every reference method defined on a subclass whose base defines the same name, the mirror's
    attribute must be defined on the mirror subclass itself, or on a class between it and the reference's base."""
    bad = []
    for modname, mod in SURFACE.items():
        m = importlib.import_module("acai_omr_amd." + modname)
        for cname, c in mod["classes"].items():
            cls = getattr(m, cname)
            for base in c["bases"]:
                bname = base.split(".")[-1]
                if bname not in mod["classes"]:
                    continue
                for mname in c["methods"]:
                    if mname in mod["classes"][bname]["methods"] and mname != "__init__":
                        basecls = getattr(m, bname)
                        if getattr(cls, mname) is getattr(basecls, mname) and (cname, mname) not in {
                                # the mirror resolves these through helper hooks of the base implementation (_stacks / _allow_pe_interpolation)
                                ("FineTuneOMREncoder", "forward"), ("FineTuneOMREncoder", "generate"), ("OMREncoder", "batchify")}:
                            bad.append(f"{cname}.{mname} resolves to {bname}.{mname}")
    assert not bad, bad


@pytest.mark.parametrize("name", ["inference", "streamed_inference"])
def test_entry_points_exist(name):
    from acai_omr_amd.inference import vitomr_inference
    assert callable(getattr(vitomr_inference, name))
