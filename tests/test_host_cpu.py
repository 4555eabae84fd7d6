"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/pp_hip.h declares,
the host-side mirrors (geometry snap, anchor tables) match the goldens taken from the reference,
and the product path refuses to run without a GPU instead of falling back."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT, golden, load_pkg

HEADER = os.path.join(ROOT, "include", "pp_hip.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    lib_mod = load_pkg("_lib")
    lib = lib_mod.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libpp_hip.so does not export {s}"
        assert s in lib_mod.PROTOTYPES, f"_lib.PROTOTYPES lacks {s}"
    assert set(lib_mod.PROTOTYPES) == set(syms)
    assert lib.pp_version() >= 1


def test_config_struct_layout_matches_header():
    lib_mod = load_pkg("_lib")
    c = lib_mod.PPConfig
    # offsets as laid out by a C compiler for the struct in pp_hip.h (natural alignment)
    assert c.voxel_size.offset == 0 and c.offset.offset == 12 and c.grid_size.offset == 24
    n = lib_mod.PP_MAX_CLASSES
    hdr = open(HEADER).read()
    assert int(re.search(r"#define PP_MAX_CLASSES (\d+)", hdr).group(1)) == n == 12
    assert c.max_voxels.offset == 36 and c.class_begin.offset == 60 and c.class_end.offset == 60 + 4 * n
    assert c.center_limit.offset == 160 and c.norm_kind.offset == 208  # 60 + 8n = 156 -> doubles aligned to 160
    assert ctypes.sizeof(c) == 232


def test_no_gpu_no_fallback(synth):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    eng = load_pkg("engine")
    with pytest.raises(RuntimeError):
        eng.Engine(synth.load_config("eight_20cm"))
    lib = load_pkg("_lib").load()
    cfg = load_pkg("_lib").PPConfig()
    assert not lib.pp_create(0, ctypes.byref(cfg))  # invalid config / no device -> NULL, message available
    assert lib.pp_last_error(None)


@pytest.mark.parametrize("name", ["eight_20cm", "ntusl_10cm", "nuscene"])
def test_snap_geometry(name, synth):
    eng = load_pkg("engine")
    g = golden(f"setup_{name}")
    vs, off, grid, rd, dr = eng.snap_geometry(synth.load_config(name))
    assert np.array_equal(vs, g["voxel_size"]) and np.array_equal(off, g["offset"]) and np.array_equal(grid, g["grid_size"])
    assert np.array_equal(rd, g["range_diff"]) and np.array_equal(dr, g["detection_range"])
    assert off.dtype == np.float32 and grid.dtype == np.int32


def test_anchor_tables_match_reference(synth):
    import hashlib
    eng = load_pkg("engine")
    g = golden("anchors_eight_20cm")
    vs, off, grid, rd, _ = eng.snap_geometry(synth.load_config("eight_20cm"))
    anchors, bv, rects, masks = eng.build_anchor_tables(off, rd, grid, vs)

    def sha(a):
        a = np.ascontiguousarray(a)
        h = hashlib.sha256()
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
        return h.hexdigest()

    assert sha(anchors) == str(g["anchors_sha"]) and sha(rects) == str(g["coors_sha"])
    assert np.array_equal(bv[g["rows"]], g["bv_rows"])
    assert list(masks.keys()) == [str(x) for x in g["class_names"]]
    assert np.array_equal(np.array(list(masks.values())), g["class_ranges"])


def test_voxel_generator_mutates_config_and_pickles(synth):
    pkg = load_pkg()
    pkg.install()
    from framework.voxel_generator import VoxelGenerator
    cfg = synth.load_config("eight_20cm")
    vg = VoxelGenerator(cfg)
    for k in ("detection_range", "detection_offset", "detection_range_diff", "grid_size"):
        assert k in cfg
    assert list(cfg["grid_size"]) == [800, 800, 1]
    vg2 = pickle.loads(pickle.dumps(vg))
    assert np.array_equal(vg2.offset, vg.offset) and vg2.max_voxels == 16000


def test_synthetic_cloud_shapes(synth):
    for name, n in (("eight_20cm", 20000), ("ntusl_10cm", 60000), ("nuscene", 34000)):
        p = synth.lidar_cloud(name, seed=3)
        assert p.shape == (n, 4) and p.dtype == np.float32 and np.isfinite(p).all()
    a, b = synth.lidar_cloud("eight_20cm", 5), synth.lidar_cloud("eight_20cm", 5)
    assert np.array_equal(a, b)
    sd = synth.seeded_state_dict(0)
    assert len([k for k in sd]) == 30 and sd["rpn.deconv3.0.weight"].shape == (256, 128, 4, 4)


def test_stateless_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation of the context-free C entry points happens before any HIP call, so it can be checked on a
    CPU-only box: every call below must return a non-zero code and must not crash."""
    import ctypes
    lib = load_pkg("_lib").load()
    i4 = (ctypes.c_int32 * 4)(0, 4, 8, 12)
    f32 = (ctypes.c_int32 * 4)(7, 7, 7, 7)
    bad_dt = (ctypes.c_int32 * 4)(7, 7, 9, 7)
    past = (ctypes.c_int32 * 4)(0, 4, 8, 14)
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    # pp_unpack_points: negative count, zero width, unknown datatype, field running past point_step, null table
    assert lib.pp_unpack_points(p, -1, 1, 16, 16, i4, f32, 0, p, None) != 0
    assert lib.pp_unpack_points(p, 4, 0, 16, 16, i4, f32, 0, p, None) != 0
    assert lib.pp_unpack_points(p, 4, 4, 64, 16, i4, bad_dt, 0, p, None) != 0
    assert lib.pp_unpack_points(p, 4, 4, 64, 16, past, f32, 0, p, None) != 0
    assert lib.pp_unpack_points(p, 4, 4, 64, 16, None, f32, 0, p, None) != 0
    assert lib.pp_unpack_points(None, 0, 4, 64, 16, i4, f32, 0, None, None) == 0  # empty cloud: nothing to do
    # pp_rotated_iou_eval: criterion outside -1..2, negative sizes; empty inputs are fine
    assert lib.pp_rotated_iou_eval(p, p, p, 1, 1, 3, None) != 0
    assert lib.pp_rotated_iou_eval(p, p, p, -1, 1, -1, None) != 0
    assert lib.pp_rotated_iou_eval(None, None, None, 0, 5, -1, None) == 0
    # pp_eval_statistics: missing outputs / inputs
    out = (ctypes.c_int64 * 3)()
    assert lib.pp_eval_statistics(None, 0, 2, 2, None, None, None, 0.5, 0.0, 0, out, None, None) != 0
    assert lib.pp_eval_statistics(None, 0, 0, 0, None, None, None, 0.5, 0.0, 0, None, None, None) != 0
    assert lib.pp_eval_statistics(None, 0, 0, 0, None, None, None, 0.5, 0.0, 0, out, None, None) == 0 and list(out) == [0, 0, 0]


def test_class_table_from_config(synth):
    """Build-side 10-class nuScenes table (no reference counterpart): engine mirror == oracle restatement, class ranges
    contiguous in table order, 20 anchors per location; the shipped 3-class configs keep the reference's hard-coded table."""
    from oracle import pp_oracle as O
    eng = load_pkg("engine")
    cfg = synth.load_config("nuscene_10class")
    names, table = eng.class_table_of(cfg)
    assert names == cfg["detect_class"] and len(names) == 10
    vs, off, grid, rd, _ = eng.snap_geometry(cfg)
    anchors, bv, rects, masks = eng.build_anchor_tables(off, rd, grid, vs, names, table)
    a = O.make_anchors(O.voxel_setup(cfg), class_table=cfg["class_table"])
    assert np.array_equal(anchors, a["anchors"]) and np.array_equal(rects, a["anchors_coors"]) and masks == a["class_masks"]
    hw = (int(grid[0]) // 2) * (int(grid[1]) // 2)
    assert anchors.shape[0] == 20 * hw and list(masks) == names
    assert [e - s for s, e in masks.values()] == [2 * hw] * 10
    assert eng.class_table_of(synth.load_config("nuscene"))[0] == ["vehicle", "pedestrian", "cyclist"]
    sd = synth.seeded_state_dict(0, num_anchor_per_loc=20)
    assert sd["heads.conv_cls.weight"].shape[0] == 20 and sd["heads.conv_box.weight"].shape[0] == 140 and sd["heads.conv_dir.bias"].shape[0] == 40


def test_stage_inputs_are_validated():
    """the stage entry points hand raw pointers to kernels, so a CPU tensor / wrong dtype / strided view /
    wrong shape must raise on the host (the reference's torch ops raise there), never reach the device."""
    import torch
    eng = load_pkg("engine")
    with pytest.raises(TypeError):
        eng._chk(torch.zeros(4, 3, dtype=torch.int32), torch.int32, (None, 3), "coors")          # CPU tensor
    with pytest.raises(TypeError):
        eng._chk(np.zeros((4, 3), np.int32), torch.int32, (None, 3), "coors")                     # not a tensor

    class FakeCuda(torch.Tensor):  # a CPU tensor that claims to be on the device: exercises the remaining checks here
        @property
        def is_cuda(self):
            return True

    t = torch.zeros(4, 3, dtype=torch.int64).as_subclass(FakeCuda)
    with pytest.raises(TypeError):
        eng._chk(t, torch.int32, (None, 3), "coors")                                              # int64 coordinates
    t = torch.zeros(3, 8, dtype=torch.int32).t().as_subclass(FakeCuda)
    with pytest.raises(ValueError):
        eng._chk(t, torch.int32, (None, 3), "coors")                                              # strided view
    t = torch.zeros(4, 4, dtype=torch.int32).as_subclass(FakeCuda)
    with pytest.raises(ValueError):
        eng._chk(t, torch.int32, (None, 3), "coors")                                              # wrong shape
    ok = torch.zeros(4, 3, dtype=torch.int32).as_subclass(FakeCuda)
    assert eng._chk(ok, torch.int32, (None, 3), "coors") is ok
