"""ROS PointCloud2 ingest (ros_node.py:55-59): device unpack against the restated
sensor_msgs.point_cloud2.read_points."""
import struct
from types import SimpleNamespace as NS

import numpy as np
import pytest

from conftest import load_pkg
from oracle import pc2_oracle as O


def make_msg(n, layout, big=False, height=1, row_pad=0, seed=0):
    """layout: list of (name, datatype, offset); point_step = end of the last field rounded up to 4 (+ tail pad)."""
    rng = np.random.default_rng(seed)
    codes = {1: 'b', 2: 'B', 3: 'h', 4: 'H', 5: 'i', 6: 'I', 7: 'f', 8: 'd'}
    sizes = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 4, 8: 8}
    step = max(o + sizes[d] for _, d, o in layout)
    step = (step + 7) // 8 * 8
    width = n // height
    row_step = width * step + row_pad
    data = bytearray(rng.integers(0, 256, row_step * height, dtype=np.uint8).tobytes())  # garbage in the padding
    e = '>' if big else '<'
    for v in range(height):
        for u in range(width):
            base = v * row_step + u * step
            for _, d, o in layout:
                if d == 7:
                    val = float(np.float32(rng.normal(0, 30)))
                elif d == 8:
                    val = float(rng.normal(0, 30))
                else:
                    lo, hi = {1: (-128, 127), 2: (0, 255), 3: (-32768, 32767), 4: (0, 65535), 5: (-2**31, 2**31 - 1), 6: (0, 2**32 - 1)}[d]
                    val = int(rng.integers(lo, hi, endpoint=True))
                struct.pack_into(e + codes[d], data, base + o, val)
    fields = [NS(name=nm, datatype=d, offset=o, count=1) for nm, d, o in layout]
    return NS(data=bytes(data), fields=fields, point_step=step, row_step=row_step, width=width, height=height, is_bigendian=big)


LAYOUTS = {
    "xyzi_f32": [("x", 7, 0), ("y", 7, 4), ("z", 7, 8), ("intensity", 7, 12)],
    "velodyne": [("x", 7, 0), ("y", 7, 4), ("z", 7, 8), ("intensity", 7, 16), ("ring", 4, 20), ("time", 8, 24)],
    "mixed": [("x", 8, 0), ("y", 7, 8), ("z", 3, 12), ("intensity", 2, 15), ("extra", 6, 16)],
    "ints": [("a", 1, 0), ("b", 4, 2), ("c", 5, 4), ("d", 6, 8), ("e", 7, 12)],
}


def test_oracle_struct_format_and_values():
    m = make_msg(5, LAYOUTS["velodyne"])
    assert O._struct_fmt(False, m.fields) == "<fffxxxxfHxxd"
    rows = list(O.read_points(m))
    assert len(rows) == 5 and len(rows[0]) == 6
    x0 = struct.unpack_from("<f", m.data, 0)[0]
    assert rows[0][0] == x0
    assert O.points_first4(m).dtype == np.float32 and O.points_first4(m).shape == (5, 4)
    assert O.points_first4(make_msg(0, LAYOUTS["xyzi_f32"])).shape == (0, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(LAYOUTS))
@pytest.mark.parametrize("big", [False, True])
def test_device_unpack_equals_read_points(name, big):
    io = load_pkg("kitti_io")
    for n, height, pad in ((1, 1, 0), (777, 1, 0), (1200, 4, 24), (20000, 1, 0)):
        m = make_msg(n, LAYOUTS[name], big=big, height=height, row_pad=pad, seed=n)
        got = io.pointcloud2_to_points(m).cpu().numpy()
        want = O.points_first4(m)
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name  # bit-exact incl. NaN payloads of garbage floats


@pytest.mark.gpu
def test_device_unpack_edge_cases():
    io = load_pkg("kitti_io")
    assert tuple(io.pointcloud2_to_points(make_msg(0, LAYOUTS["xyzi_f32"])).shape) == (0, 4)
    m = make_msg(8, LAYOUTS["xyzi_f32"])
    m.fields = m.fields[:3]
    with pytest.raises(ValueError):
        io.pointcloud2_to_points(m)
    m = make_msg(8, LAYOUTS["xyzi_f32"])
    m.data = m.data[:-1]
    with pytest.raises(ValueError):
        io.pointcloud2_to_points(m)
