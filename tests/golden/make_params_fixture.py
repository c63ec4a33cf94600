"""Writes tests/golden/gluon_tiny.params BYTE BY BYTE from the published MXNet 1.x NDArray-list layout
(src/ndarray/ndarray.cc NDArray::Save, src/c_api/c_api.cc MXNDArraySave) - deliberately without
tennis_amd.params_io, so that the reader is pinned against something other than its own writer.  The content is
what ``mx.gluon.Block.save_parameters`` of a ``TemporalPooling(model=None, num_classes=3)``-shaped block writes
(reference models/vision/definitions.py:60-61: one Dense ``classes``): structural names ``classes.weight`` /
``classes.bias``, plus one float16 array and one ``aux:``-prefixed name as Module checkpoints carry them.

A second file, gluon_gnmt_tiny.params, carries the structural names ``save_parameters`` writes for the captioner the
reference builds (train_gnmt.py:221-229: gluonnlp NMTModel around models/captioning/gnmt.py:84-111,212-221 with
cell_type gru, num_layers 2, num_bi_layers 1, hidden 2, 3-d source features, embed 2, vocabulary 5): the
``rnn_cells`` HybridSequential children by index, ``l_cell`` / ``r_cell`` of the BidirectionalCell, the attention
cell's single projection, ``tgt_embed.0`` (HybridSequential(Embedding, Dropout)) and the ``tgt_proj`` Dense.  Array
values are ``index_of_array + 0.01 * position`` so that every mapping and the one transposition are checkable.

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


# ---- captioner checkpoint ------------------------------------------------------------------------------------
H, F, E, V, G = 2, 3, 2, 5, 3
cell = lambda fin: [("i2h_weight", (G * H, fin)), ("h2h_weight", (G * H, H)), ("i2h_bias", (G * H,)), ("h2h_bias", (G * H,))]
entries = []
entries += [("encoder.rnn_cells.0.l_cell." + n, sh) for n, sh in cell(F)]
entries += [("encoder.rnn_cells.0.r_cell." + n, sh) for n, sh in cell(F)]
entries += [("encoder.rnn_cells.1." + n, sh) for n, sh in cell(2 * H)]
entries += [("decoder.attention_cell._proj_query.weight", (H, H))]
entries += [("decoder.rnn_cells.0." + n, sh) for n, sh in cell(E + H)]
entries += [("decoder.rnn_cells.1." + n, sh) for n, sh in cell(2 * H)]
entries += [("tgt_embed.0.weight", (V, E)), ("tgt_proj.weight", (V, H)), ("tgt_proj.bias", (V,))]
out = bytearray()
out += struct.pack("<QQQ", 0x112, 0, len(entries))
for i, (name, shape) in enumerate(entries):
    n = 1
    for d in shape:
        n *= d
    out += ndarray_v2(shape, 0, struct.pack("<%df" % n, *[i + 0.01 * j for j in range(n)]))
out += struct.pack("<Q", len(entries))
for name, _ in entries:
    out += struct.pack("<Q", len(name)) + name.encode()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gluon_gnmt_tiny.params")
with open(path, "wb") as f:
    f.write(bytes(out))
print(path, len(out), "bytes")
