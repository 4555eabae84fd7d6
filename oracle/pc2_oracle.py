"""CPU restatement of sensor_msgs.point_cloud2.read_points as the reference's ROS callback uses it
(ros_node.py:55-59) -- TEST INFRASTRUCTURE ONLY.

The library is a third-party dependency that is not in /root/reference (ROS `sensor_msgs`, common_msgs 1.13.x,
sensor_msgs/point_cloud2.py): `read_points(cloud)` builds one `struct` format from the fields sorted by offset
(`_get_struct_fmt`: '>' or '<', pad bytes 'x' up to each field's offset, then the field's type code from
`_DATATYPES` = {1:'b', 2:'B', 3:'h', 4:'H', 5:'i', 6:'I', 7:'f', 8:'d'}) and yields `unpack_from(data, offset)` for
every (row v, column u) at offset `row_step * v + point_step * u`.  The reference then keeps the first four values
of each tuple and casts to float32.  Parity is anchored on that call site; there is no golden vector in the
reference for it ("parity unpinned" for this function: the oracle is the published algorithm)."""
import struct

import numpy as np

_DATATYPES = {1: ('b', 1), 2: ('B', 1), 3: ('h', 2), 4: ('H', 2), 5: ('i', 4), 6: ('I', 4), 7: ('f', 4), 8: ('d', 8)}


def _struct_fmt(is_bigendian, fields):
    fmt = '>' if is_bigendian else '<'
    offset = 0
    for f in sorted(fields, key=lambda f: f.offset):
        if offset < f.offset:
            fmt += 'x' * (f.offset - offset)
            offset = f.offset
        code, size = _DATATYPES[f.datatype]
        fmt += code * f.count
        offset += f.count * size
    return fmt


def read_points(msg):
    fmt = _struct_fmt(msg.is_bigendian, msg.fields)
    unpack = struct.Struct(fmt).unpack_from
    data = bytes(msg.data)
    for v in range(msg.height):
        off = msg.row_step * v
        for _ in range(msg.width):
            yield unpack(data, off)
            off += msg.point_step


def points_first4(msg):
    """ros_node.py:57: np.asarray(list(pc2.read_points(msg)))[:, :4].astype(np.float32)."""
    rows = list(read_points(msg))
    if not rows:
        return np.zeros((0, 4), np.float32)
    return np.asarray(rows)[:, :4].astype(np.float32)
