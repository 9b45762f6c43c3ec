"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol include/gsrast.h
declares, validates arguments without touching a GPU, and the drop-in Python surface has the
reference's shape (gaussian_renderer/__init__.py:36-51,85-93)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from diff_gaussian_rasterization import _native
    if not os.path.exists(_native.lib_path()):
        _native.build()
    return _native


def test_header_symbols_are_exported(native):
    hdr = open(os.path.join(ROOT, "include", "gsrast.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char \*)\s*\**(gsr_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert {"gsr_forward_preprocess", "gsr_forward_render", "gsr_backward_render", "gsr_backward_geom",
            "gsr_mark_visible", "gsr_workspace_sizes", "gsr_binning_size", "gsr_version", "gsr_last_error"} <= declared
    lib = native.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gsrast.h but not exported"
    assert set(native.EXPORTS) == declared
    assert lib.gsr_version() == 12


def test_argument_validation_without_gpu(native):
    lib = native.load()
    bad = native.make_desc(10, 5, 16, 64, 64, 0.5, 0.5, 1.0, False, False)
    g, i = C.c_size_t(0), C.c_size_t(0)
    assert lib.gsr_workspace_sizes(C.byref(bad), C.byref(g), C.byref(i)) == -1
    assert b"sh_degree" in lib.gsr_last_error()
    ok = native.make_desc(1000, 3, 16, 100, 60, 0.5, 0.5, 1.0, False, False)
    gb, ib = native.workspace_sizes(ok)
    assert gb >= 1000 * 57 and ib >= 100 * 60 * 8 + 7 * 4 * 8
    # six u32 and one flag byte per instance, one 4 KB checkpoint per segment of them, the blend backward's unit lists
    seg = native.bwd_segment_entries()
    assert 5000 * 25 + (5000 // seg) * 4096 <= native.binning_size(ok, 5000) < 5000 * 25 + (5000 // seg + 2) * 4096 + 64 * (5000 // seg + 1 + 16 * 8 * 5) + 4096
    # backward-only gradient rows: 48 B per EMITTED instance; the emission bound when the count stayed on the device
    plan = native.FramePlan(); plan.instances_emitted = 1234
    assert 1234 * 48 <= native.backward_rows_size(ok, plan) < 1234 * 48 + 512
    plan.instances_emitted = -1; plan.chunks_run = 2; plan.chunk_instances_max[0] = 1000; plan.chunk_instances_max[1] = 500
    assert 1500 * 48 <= native.backward_rows_size(ok, plan) < 1500 * 48 + 512
    with pytest.raises(native.GsrError, match="sh_coeffs"):
        native.workspace_sizes(native.make_desc(10, 1, 99, 64, 64, 0.5, 0.5, 1.0, False, False))
    # exactly-one-of rules are enforced at the ABI too (NULL device pointers are never dereferenced here)
    cam = native.Camera(1, 1, 1, 1)
    gs = native.Gaussians(1, None, None, 1, 1, 1, None)
    plan = native.FramePlan()
    rc = lib.gsr_forward_preprocess(C.byref(ok), C.byref(cam), C.byref(gs), C.c_void_p(1), None, C.c_void_p(1), C.byref(plan), None)
    assert rc == -1 and b"shs / colors_precomp" in lib.gsr_last_error()


def test_python_surface_matches_reference_call_site():
    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    rs = dgr.GaussianRasterizationSettings(8, 8, .5, .5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    r = dgr.GaussianRasterizer(raster_settings=rs)
    z = torch.zeros(4, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=z, means2D=z, opacities=torch.zeros(4, 1), scales=z, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair"):
        r(means3D=z, means2D=z, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 1, 3))
    # CPU tensors: the product has no CPU path and says so
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=z, means2D=z, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 1, 3), scales=z, rotations=torch.zeros(4, 4))
    assert hasattr(r, "markVisible")
    import gaussian_renderer
    assert callable(gaussian_renderer.render)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "structured-gaussian-splatting_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libgsr_oracle" not in src, f


def test_python_mirror_of_binning_size_matches_the_library():
    """diff_gaussian_rasterization._binning_bytes / the first-chunk capacity rule are evaluated in Python between the plan
    readback and the first launch of stage 2 (no ctypes round trips while the stream idles): they must equal the C ABI's."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    desc = N.make_desc(1000, 3, 16, 640, 480, 0.5, 0.5, 1.0, False, False)
    for n in (0, 1, 63, 64, 65, 1000, 123_457, 28_201_899, 492_421_683):
        assert dgr._binning_bytes(n, 40 * 30) == N.binning_size(desc, n), n
    plan = N.FramePlan()
    for chunks, first, R in ((1, 500, 500), (3, 2_000_000, 28_000_000), (2, 900_000, 1_000_000), (4, 10, 5_000_000_000)):
        plan.num_chunks, plan.num_rendered = chunks, R
        plan.chunk_instances_max[0] = first
        want = min(first + first // 4 + (1 << 20), R) if chunks > 1 else R
        assert N.binning_first_chunk_capacity(plan) == want


def test_ctypes_mirrors_have_the_layout_of_the_header(native, tmp_path):
    """include/gsrast.h is plain C: gcc compiles it, and size + every field offset of each struct equals the ctypes
    mirror's (the layout a cgo / JNI / ctypes binding of the header would see)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not on PATH")
    pairs = (("gsr_frame_desc", native.FrameDesc), ("gsr_frame_plan", native.FramePlan), ("gsr_camera", native.Camera),
             ("gsr_gaussians", native.Gaussians), ("gsr_grads", native.Grads), ("gsr_debug_views", native.DebugViews))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gsrast.h"', 'int main(void) {']
    for cname, mirror in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in mirror._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, mirror in pairs:
        assert int(got[cname]) == C.sizeof(mirror), cname
        for fname, _ in mirror._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(mirror, fname).offset, f"{cname}.{fname}"
