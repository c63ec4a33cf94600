"""Writes tests/golden/gluon_tiny.params BYTE BY BYTE from the published MXNet 1.x NDArray-list layout
(src/ndarray/ndarray.cc NDArray::Save, src/c_api/c_api.cc MXNDArraySave) - deliberately without
tennis_amd.params_io, so that the reader is pinned against something other than its own writer.  The content is
what ``mx.gluon.Block.save_parameters`` of a ``TemporalPooling(model=None, num_classes=3)``-shaped block writes
(reference models/vision/definitions.py:60-61: one Dense ``classes``): structural names ``classes.weight`` /
``classes.bias``, plus one float16 array and one ``aux:``-prefixed name as Module checkpoints carry them.

    python tests/golden/make_params_fixture.py
"""
import os
import struct

out = bytearray()
out += struct.pack("<Q", 0x112)            # kMXAPINDArrayListMagic
out += struct.pack("<Q", 0)                # reserved
out += struct.pack("<Q", 4)                # number of NDArrays


def ndarray_v2(shape, dtype_flag, payload):
    b = bytearray()
    b += struct.pack("<I", 0xF993FAC9)     # NDARRAY_V2_MAGIC
    b += struct.pack("<i", 0)              # kDefaultStorage
    b += struct.pack("<I", len(shape))     # TShape: ndim (uint32) ...
    for d in shape:
        b += struct.pack("<q", d)          # ... and int64 dims
    b += struct.pack("<ii", 1, 0)          # Context: cpu(0)
    b += struct.pack("<i", dtype_flag)     # mshadow type flag
    b += payload
    return b


w = [0.5, -1.25, 2.0, 0.125, 3.5, -0.75]                                   # (2, 3) float32, row-major
out += ndarray_v2((2, 3), 0, struct.pack("<6f", *w))
out += ndarray_v2((2,), 0, struct.pack("<2f", 0.25, -0.5))
out += ndarray_v2((3,), 2, struct.pack("<3e", 1.0, -2.0, 0.5))            # float16 (flag 2)
out += ndarray_v2((1,), 0, struct.pack("<1f", 7.0))
names = [b"classes.weight", b"classes.bias", b"half_vector", b"aux:running_thing"]
out += struct.pack("<Q", len(names))
for n in names:
    out += struct.pack("<Q", len(n)) + n

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gluon_tiny.params")
with open(path, "wb") as f:
    f.write(bytes(out))
print(path, len(out), "bytes")
