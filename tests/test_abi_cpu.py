"""The C-ABI shared library loads without a GPU and exports every symbol include/robir_hip.h declares;
host-side logic that needs no device (packing plans, synthetic data, conf accessors)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(name):
    hdr = open(os.path.join(ROOT, "include", name)).read()
    return sorted(set(re.findall(r"^(?:int|long|const char\*) (rb_[a-z0-9_]+)\s*\(", hdr, re.M)))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if " T rb_" in l)


def test_library_exports_header_symbols():
    """The default library exports exactly what include/robir_hip.h declares (ABI version 8: 89 entry points -- the retired kernel
    generations left it in round 5, the seven helper kernels of csrc/surface.hip and rb_cesr_net_f16_points joined in round 6), the legacy library -- where it has
    been built: ROBIR_BUILD_LEGACY=1, optional since round 6 -- exactly that plus include/robir_hip_legacy.h."""
    from robir_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build(legacy=False)
    L = _lib.lib()
    assert L.rb_abi_version() == 8
    syms, leg = _header_symbols("robir_hip.h"), _header_symbols("robir_hip_legacy.h")
    assert 40 <= len(syms) <= 100 and len(leg) >= 30 and not set(syms) & set(leg)
    assert _exported(_lib.LIB_PATH) == syms                                      # nothing missing, nothing undeclared, nothing legacy
    assert L.rb_packed_layer_floats(256, 256) == 16 * (16 + 256 * 16)
    assert _lib.resolve("rb_dvis_fused_x6t")[0] is L
    if os.path.exists(_lib.LEGACY_PATH):
        assert _exported(_lib.LEGACY_PATH) == sorted(syms + leg)                 # the superset
        assert _lib.legacy().rb_abi_version() == 8
        # a retired entry point resolves to the legacy library, a current one to the default library
        assert _lib.resolve("rb_dvis_fused_v2")[0] is _lib.legacy()


def test_error_reporting_without_gpu():
    """Argument validation happens before any launch: a null pointer returns non-zero and sets rb_last_error()."""
    import ctypes
    from robir_amd import _lib
    L = _lib.lib()
    rc = L.rb_vis_mlp_points(ctypes.c_void_p(0), ctypes.c_void_p(0), ctypes.c_long(8), ctypes.c_int(1), ctypes.c_void_p(0), ctypes.c_void_p(0),
                             ctypes.c_void_p(0))
    assert rc != 0 and b"null pointer" in L.rb_last_error()
    assert L.rb_vis_mlp_points(ctypes.c_void_p(0), ctypes.c_void_p(0), ctypes.c_long(0), ctypes.c_int(1), ctypes.c_void_p(0), ctypes.c_void_p(0),
                               ctypes.c_void_p(0)) == 0


def test_no_oracle_import_in_product():
    """The product package must never import the oracle (test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "robir_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "robir_oracle" not in src and "ref_shim" not in src, f


def test_model_state_dict_matches_reference_keys(synth_weights):
    import torch
    from robir_amd import renderer
    m = renderer.IDRNetwork(renderer.hotdog_conf())
    r = m.load_state_dict({k: torch.from_numpy(v) for k, v in synth_weights.items()}, strict=False)
    assert not r.missing_keys and not r.unexpected_keys
    assert len(m.state_dict()) == 134            # the reference's IDRNetwork has exactly these 134 entries (golden gen)


def test_synth_determinism():
    from robir_amd import synth
    a, b = synth.synth_state_dict(0), synth.synth_state_dict(0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    d1, d2 = synth.pbr_draws(3, 17, chunk_id=5), synth.pbr_draws(3, 17, chunk_id=5)
    assert all(np.array_equal(d1[k], d2[k]) for k in d1)
    assert d1["dvis_theta"].shape == (128, 32) and d1["svis_phi_ind"].shape == (17, 8)


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from robir_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RobirHipError):
        _lib.lib()
    monkeypatch.setattr(_lib, "_legacy", None)
    monkeypatch.setattr(_lib, "LEGACY_PATH", str(tmp_path / "nope_legacy.so"))
    with pytest.raises(_lib.RobirHipError, match="LEGACY library"):
        _lib.legacy()


def test_mlp_precision_switch(monkeypatch):
    """ROBIR_MLP_PRECISION selects the arithmetic of the stand-alone MLP kernels; anything else is an error, not a guess."""
    from robir_amd import nets
    from robir_amd import precision
    monkeypatch.delenv("ROBIR_MLP_PRECISION", raising=False)
    monkeypatch.delenv("ROBIR_VIS_PRECISION", raising=False)
    monkeypatch.delenv("ROBIR_PRECISION", raising=False)
    assert nets.mlp_precision() == "f16x6" and precision.vis_precision() == "f16x6"     # default policy: not narrower than fp32
    monkeypatch.setenv("ROBIR_PRECISION", "split")
    assert nets.mlp_precision() == "f16x3" and precision.vis_precision() == "f16x3-auto"
    monkeypatch.setenv("ROBIR_MLP_PRECISION", "fp32")
    assert nets.mlp_precision() == "fp32"
    monkeypatch.setenv("ROBIR_MLP_PRECISION", "bf16")
    import pytest
    with pytest.raises(ValueError):
        nets.mlp_precision()
    # the whole policy table (robir_amd/precision.py): light-visibility kernel, stand-alone MLPs, the two CESR nets
    monkeypatch.delenv("ROBIR_MLP_PRECISION")
    monkeypatch.delenv("ROBIR_CESR_PRECISION", raising=False)
    want = {"exact": ("f16x6", "f16x6", "f16x6"), "split": ("f16x3-auto", "f16x3", "f16x3"),
            "f16": ("f16x1", "f16x3", "f16x1"),            # the labelled throughput policy: plain f16 where a one-product kernel exists
            "f16-vis": ("f16x1", "f16x6", "f16x6")}        # round 4's meaning of f16: the light-visibility MLP only
    for pol, (vis, mlp, cesr) in want.items():
        monkeypatch.setenv("ROBIR_PRECISION", pol)
        assert (precision.vis_precision(), precision.mlp_precision(), precision.cesr_precision()) == (vis, mlp, cesr), pol
    monkeypatch.setenv("ROBIR_PRECISION", "f16")
    monkeypatch.setenv("ROBIR_CESR_PRECISION", "f16x6")     # the CESR nets alone back on exact operands
    assert precision.cesr_precision() == "f16x6" and precision.vis_precision() == "f16x1"
    monkeypatch.delenv("ROBIR_CESR_PRECISION")
    monkeypatch.setenv("ROBIR_MLP_PRECISION", "f16x6")      # an explicit MLP override also covers the CESR nets
    assert precision.cesr_precision() == "f16x6"
    monkeypatch.setenv("ROBIR_PRECISION", "fp16")
    with pytest.raises(ValueError):
        precision.policy()


def test_object_surface_of_the_reference_model():
    """Attributes the reference's runners touch on the model (SURVEY 8b 'Object surface'): present, callable where the
    runners call them, `get_sg_render` assignable per instance (train_pbr.py:413)."""
    import types
    from robir_amd import renderer
    m = renderer.IDRNetwork(renderer.hotdog_conf())
    for path in ("implicit_network.gradient", "implicit_network.batch_borrow_color", "implicit_network.neus_model.dev",
                 "ray_tracer.generate", "octree_ray_tracer.generate", "indirect_illum_network", "visibility_network",
                 "envmap_material_network.get_light", "envmap_material_network.load_light", "envmap_material_network.lgtSGs",
                 "envmap_material_network.specular_reflectance", "envmap_material_network.upper_hemi",
                 "envmap_material_network.spec_brdf_encoder_layer.var", "envmap_material_network.spec_brdf_encoder_layer.lc_act",
                 "gamma.hdr_shift.as_input", "gamma.hdr_shift.hdr2ldr", "gamma.hdr_shift.ldr2hdr", "gamma.hdr_shift.fit_data",
                 "get_idr_render", "get_sg_render", "trace_radiance", "state_dict", "load_state_dict"):
        obj = m
        for part in path.split("."):
            assert hasattr(obj, part), path
            obj = getattr(obj, part)
    assert hasattr(m.envmap_material_network, "envmap")
    hook = lambda self, *a, **k: "hooked"
    m.get_sg_render = types.MethodType(hook, m)
    assert m.get_sg_render() == "hooked"
    assert m.octree_ray_tracer.max_iter == 32


def test_overlay_exports_reference_names():
    """overlay/model/*.py (PEP-420 namespace overlay put in front of the reference's model/ directory) re-export the public
    names the reference's runners and scripts import from those modules."""
    import importlib
    import sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "overlay")
    sys.path.insert(0, root)
    try:
        want = {
            "model.sg_render": ["render_with_all_sg", "render_with_sg", "compute_envmap", "render_envmap", "render_envmap_sg",
                                "get_diffuse_visibility", "get_specular_visibility", "norm_axis", "TINY_NUMBER"],
            "model.implicit_differentiable_renderer": ["IDRNetwork", "IndirctIllumNetwork", "VisNetwork"],
            "model.octree_tracing": ["OctreeTracing", "OctreeVisModel"],
            "model.ray_tracing": ["RayTracing"],
            "model.sg_envmap_material": ["SparseAE", "EnvmapMaterialNetwork", "fibonacci_sphere", "compute_energy"],
            "model.neus_model": ["SDFNetwork", "RenderingNetwork", "SingleVarianceNetwork", "NeuSModel", "ImplicitNetworkMy"],
            "model.embedder": ["get_embedder"],
            "model.color_correction": ["ACESToneMapping", "GammaCorrect"],
            "model.sdf_render": ["render_neus", "Rays"],
        }
        for mod, names in want.items():
            m = importlib.import_module(mod)
            for n in names:
                assert hasattr(m, n), (mod, n)
    finally:
        sys.path.remove(root)
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="needs the reference checkout (build container only)")
def test_overlay_shadows_reference_modules_by_path_order():
    """INTEGRATION.md seam B: with overlay/ ahead of the reference on the path, `model.<overlaid>` comes from this repo and
    every other `model.*` module still comes from the reference (PEP-420: `model/` has no __init__.py on either side).
    Run in a subprocess so that the module cache of the test session stays clean."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, importlib.util\n"
        "a = importlib.util.find_spec('model.sg_render').origin\n"
        "b = importlib.util.find_spec('model.octree_tracing').origin\n"
        "c = importlib.util.find_spec('model.loss').origin\n"
        "print(a); print(b); print(c)\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "overlay"), root, "/root/reference"]))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    a, b, c = out.stdout.strip().splitlines()[-3:]
    assert a.startswith(os.path.join(root, "overlay")) and b.startswith(os.path.join(root, "overlay"))
    assert c.startswith("/root/reference/model")


def test_forward_only_guard():
    """The kernels have no backward: a training-mode forward with grad enabled raises instead of returning detached tensors
    (checked before any kernel is launched, so this runs without a GPU)."""
    import warnings
    import torch
    from robir_amd import nets, renderer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = renderer.IDRNetwork(renderer.hotdog_conf())
    x = torch.zeros(4, 3)
    inp = {"uv": torch.zeros(1, 4, 2), "pose": torch.eye(4)[None], "intrinsics": torch.eye(3)[None],
           "object_mask": torch.ones(1, 4, dtype=torch.bool)}
    with torch.enable_grad():
        m.train()
        for call in (lambda: m(inp, trainstage="Material"), lambda: m.visibility_network(x, x),
                     lambda: m.implicit_network(x), lambda: m.implicit_network.gradient(x),
                     lambda: m.envmap_material_network(x, train_spec=True), lambda: m.indirect_illum_network(x, x[:, :1]),
                     lambda: m.trace_radiance({"points": x, "hdr_shift": x[:, :1], "network_object_mask": x[:, 0] > 1})):
            with pytest.raises(nets.ForwardOnlyError):
                call()
        # frozen parameters, eval mode or no_grad are all fine for the guard (the call then proceeds to the kernels,
        # which need the GPU: only the guard itself is exercised here)
        nets.forward_only_guard(m.eval())
        m.train()
        with torch.no_grad():
            nets.forward_only_guard(m)
        for p in m.parameters():
            p.requires_grad_(False)
        nets.forward_only_guard(m)


def test_no_compiler_written_m0_in_the_lds_dma_kernels():
    """The two-tile exact-operand kernels issue the LDS-DMA pieces of a chunk behind ONE M0 write (x6t_engine.h: xt_copy_piece_seq); that is
    sound only while the compiler writes M0 nowhere in those kernels.  tools/check_m0.py disassembles both libraries and fails on any M0
    write that is not the inline-assembly `s_mov_b32 m0, sN ; s_nop 0 ; global_load_lds_*` pattern."""
    import importlib.util
    from robir_amd import _lib
    spec = importlib.util.spec_from_file_location("check_m0", os.path.join(ROOT, "tools", "check_m0.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    for lib in (_lib.LIB_PATH, _lib.LEGACY_PATH):
        kernels, writes, bad = mod.check(lib)
        assert kernels >= 15 and writes > 1000 and not bad, (lib, kernels, writes, bad[:5])
