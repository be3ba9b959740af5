"""Dump the PUBLIC SURFACE of the nine reference modules the overlay shadows as JSON: for every public
function / class defined in the module, its parameters (name, kind, repr of the default) and, for classes, the same for every public
method plus __init__ / __call__, and the base-class names.  TEST INFRASTRUCTURE (build container only): an interface description --
names and parameter lists, no source text -- committed as tests/golden/reference_surface.json and checked by tests/test_surface_cpu.py,
which re-runs this dump when /root/reference is present and asserts the committed file is current.

    python oracle/dump_reference_surface.py [out.json]
"""
import importlib
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MODULES = ["sg_render", "implicit_differentiable_renderer", "neus_model", "octree_tracing", "sg_envmap_material", "embedder",
           "color_correction", "sdf_render", "ray_tracing"]


def default_repr(v):
    if v is inspect.Parameter.empty:
        return None
    if inspect.isclass(v) or inspect.isroutine(v):
        return "<" + getattr(v, "__name__", repr(v)) + ">"       # classes / builtins by NAME (their repr carries module paths / addresses)
    return repr(v)


def params(fn):
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return None
    return [[p.name, p.kind.name, default_repr(p.default)] for p in sig.parameters.values()]


def surface(mod, modname):
    out = {"functions": {}, "classes": {}, "constants": {}}
    for name, obj in vars(mod).items():
        if name.startswith("_"):
            continue
        if isinstance(obj, (int, float, str)) and not isinstance(obj, bool):
            out["constants"][name] = obj
            continue
        if not (inspect.isfunction(obj) or inspect.isclass(obj)) or getattr(obj, "__module__", None) != modname:
            continue
        if inspect.isfunction(obj):
            out["functions"][name] = params(obj)
            continue
        methods = {}
        for k, v in vars(obj).items():
            if k.startswith("_") and k not in ("__init__", "__call__"):
                continue
            if isinstance(v, (staticmethod, classmethod)):
                methods[k] = {"kind": type(v).__name__, "params": params(v.__func__)}
            elif inspect.isfunction(v):
                methods[k] = {"kind": "method", "params": params(v)}
        out["classes"][name] = {"bases": [b.__name__ for b in obj.__mro__[1:-1]], "methods": methods}
    return out


def dump(prefix="model."):
    return {m: surface(importlib.import_module(prefix + m), prefix + m) for m in MODULES}


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    import ref_shim
    ref_shim.install()
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(HERE), "tests", "golden", "reference_surface.json")
    json.dump(dump(), open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)
