"""Host-side mirror of the reference API (CPU-only checks; no compute calls)."""
import ctypes
import os
import re

import pytest
import torch

from tests import helpers as H


def test_state_dict_matches_reference_manifest():
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    m = PlanningModel(radius=120)
    sd = m.state_dict()
    man = H.manifest()
    assert list(sd.keys()) == list(man.keys())          # names AND order
    for k, v in sd.items():
        assert list(v.shape) == man[k], k
    assert sum(p.numel() for p in m.parameters()) == 4240589
    # perturbed fixture weights load strictly
    m.load_state_dict(H.weights(), strict=True)


def test_model_refuses_cpu_forward():
    from rift_amd.planning.pluto.model.pluto_model import PlanningModel
    m = PlanningModel(radius=120)
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        m.forward(H.build_batch("small")["cur_pluto_feature_torch"])


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on a GPU-less box and exports every function include/rift_hip.h declares."""
    from rift_amd import _ffi, build
    build.build()
    lib = _ffi.load_library()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rift_hip.h")).read()
    declared = set(re.findall(r"\b(rift_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in rift_hip.h but not exported"
    assert set(_ffi.EXPORTS) == declared


def test_trampolines_are_generated_from_the_header():
    """csrc/abi.cpp / abi_list.h (the exported symbols: routing onto the bf16- or fp16-operand build that owns a context) are what
    tools/gen/abi_trampolines.py derives from include/rift_hip.h -- the header stays the single statement of the interface."""
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("abi_trampolines", os.path.join(repo, "tools", "gen", "abi_trampolines.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lst, cpp = mod.render()
    assert open(mod.OUT_LIST).read() == lst and open(mod.OUT_CPP).read() == cpp, "run python tools/gen/abi_trampolines.py"
    from rift_amd import _ffi
    lib = _ffi.load_library()
    for sym in ("rift_vtable_bf16", "rift_vtable_fp16"):          # one table per operand-format build
        assert hasattr(lib, sym), sym
    # no GPU here: creating a context for an unknown operand format is refused before any device call
    ctx = ctypes.c_void_p()
    assert lib.rift_ctx_create_ex(0, 7, ctypes.byref(ctx)) == -1


def test_bench_kernel_labels_resolve_in_the_committed_counter_tables():
    """bench.py's roofline object reads HBM traffic and MFMA counters of the dominant kernel from the newest committed PMC passes
    (profiles/rNN_pmc_*): every kernel the newest committed bench line lists must resolve there, so that a renamed kernel shows up here
    and not as a silent `traffic: null` at round end."""
    import glob
    import importlib.util
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lines = sorted(glob.glob(os.path.join(repo, "profiles", "r*_bench.json")))
    assert lines and bench.pmc_traffic_file() is not None
    assert os.path.basename(lines[-1])[:3] == os.path.basename(bench.pmc_traffic_file())[:3], "bench line and traffic table of different rounds"
    roof = json.load(open(lines[-1]))["roofline"]
    for label in roof["per_kernel_ms_per_step"]:
        assert bench.pmc_traffic(label) is not None, f"{label}: no traffic entry in {bench.pmc_traffic_file()}"
    big = [k for k, ms in roof["per_kernel_ms_per_step"].items() if ms > 0.05]
    for label in big:
        assert bench.pmc_mfma(label) is not None, f"{label}: no MFMA counters in the committed pass"


def test_struct_layouts_match_header_sizes():
    from rift_amd import _ffi
    # 6 int32 + 26 pointers + 1 int32 (padded) ; 5 pointers ; 8 pointers + 2 floats ; 11 pointers
    assert ctypes.sizeof(_ffi.RiftFeatureBatch) == 24 + 26 * 8 + 8
    assert ctypes.sizeof(_ffi.RiftOutputs) == 40
    assert ctypes.sizeof(_ffi.RiftLossIn) == 72
    assert ctypes.sizeof(_ffi.RiftLossOut) == 88   # 11 pointers (incl. the optional f64 exchange buffer)
    assert ctypes.sizeof(_ffi.RiftTensorDesc) == 8 + 8 + 8 + 8 + 32


def test_ctypes_structs_match_the_header_as_a_c_compiler_lays_it_out(tmp_path):
    """include/rift_hip.h compiled by gcc as C: sizeof and every field offset of the structs the Python binding mirrors (a ctypes Structure
    that drifts from the header corrupts arguments silently)."""
    import shutil
    import subprocess
    from rift_amd import _ffi
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"RiftTickCBV": _ffi.RiftTickCBV, "RiftRolloutIO": _ffi.RiftRolloutIO, "RiftFeatureBatch": _ffi.RiftFeatureBatch,
               "RiftOutputs": _ffi.RiftOutputs, "RiftLossIn": _ffi.RiftLossIn, "RiftLossOut": _ffi.RiftLossOut}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rift_hip.h"', 'int main(void) {']
    for name, st in structs.items():
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for field, *_ in st._fields_:
            lines.append(f'  printf(" %zu", offsetof({name}, {field}));')
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(repo, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for line in out:
        name, size, *offs = line.split()
        st = structs[name]
        assert ctypes.sizeof(st) == int(size), (name, ctypes.sizeof(st), size)
        assert [getattr(st, f).offset for f, *_ in st._fields_] == [int(o) for o in offs], name


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under rift_amd/ may import or reference it."""
    import pathlib
    repo = pathlib.Path(__file__).resolve().parents[1]
    offenders = []
    for root in (repo / "rift_amd", repo / "tools"):          # tools/: profiling / micro-benchmarks of the product, same rule
        for f in list(root.rglob("*.py")) + list(root.rglob("*.h")) + list(root.rglob("*.hip")) + list(root.rglob("*.sh")):
            txt = f.read_text()
            if "import oracle" in txt or "from oracle" in txt or "oracle/" in txt:
                offenders.append(str(f))
    assert not offenders, offenders
    # bench.py and __graft_entry__.py may use it only inside cpu_baseline() / smoke()
    import ast
    for name, allowed in (("bench.py", {"cpu_baseline"}), ("__graft_entry__.py", {"smoke"})):
        tree = ast.parse((repo / name).read_text())
        for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
            for node in (fn.body if isinstance(fn, ast.Module) else []):
                if isinstance(node, (ast.Import, ast.ImportFrom)):
                    mod = node.module if isinstance(node, ast.ImportFrom) else ",".join(a.name for a in node.names)
                    assert "oracle" not in (mod or ""), f"{name}: module-level oracle import"
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            uses = any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ((n.module if isinstance(n, ast.ImportFrom) else ",".join(a.name for a in n.names)) or "")
                       for n in ast.walk(fn))
            assert not uses or fn.name in allowed, f"{name}: {fn.name} imports the oracle"


def test_scene_dump_round_trip(tmp_path):
    """save_scenes / load_scenes (the dump format BASELINE config 0's "pre-dumped rollout scenes" needs): 8 ragged scenes come back with
    identical keys, dtypes, shapes and values, and collate to the same padded batch."""
    from rift_amd import synthetic as syn
    from rift_amd.replay import load_scenes, save_scenes
    scenes = [syn.make_scene(900 + i, num_agents=10 + i, num_polygons=6 + (i % 3), r_min=1, r_max=4) for i in range(8)]
    path = str(tmp_path / "replay.npz")
    save_scenes(path, scenes)
    back = load_scenes(path)
    assert len(back) == 8

    def same(a, b, where):
        if isinstance(a, dict):
            assert set(a) == set(b), where
            for k in a:
                same(a[k], b[k], f"{where}/{k}")
        else:
            a, b = torch.as_tensor(a), torch.as_tensor(b)
            assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), where

    for i, (s, t) in enumerate(zip(scenes, back)):
        same(s["feature"], t["feature"], f"scene {i} feature")
        same(s["extras"], t["extras"], f"scene {i} extras")
    b0 = syn.collate_features([s["feature"] for s in scenes])
    b1 = syn.collate_features([s["feature"] for s in back])
    same(b0, b1, "collated batch")
    with pytest.raises(ValueError):
        import numpy as np
        np.savez(str(tmp_path / "other.npz"), x=np.zeros(3))
        load_scenes(str(tmp_path / "other.npz"))
