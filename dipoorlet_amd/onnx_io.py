"""Minimal ONNX (protobuf wire format) reader / writer — no `onnx` or `protobuf` package needed.

The reference loads models with `onnx.load` (dipoorlet/__main__.py:95-98); that package is not
installable here, and the calibration path only needs a small part of the schema: graph topology,
initializers, value infos, node attributes.  Field numbers follow onnx.proto (ONNX IR v7+):

  ModelProto   ir_version=1 producer_name=2 graph=7 opset_import=8
  OperatorSetIdProto domain=1 version=2
  GraphProto   node=1 name=2 initializer=5 input=11 output=12 value_info=13
  NodeProto    input=1 output=2 name=3 op_type=4 attribute=5 domain=7
  AttributeProto name=1 f=2 i=3 s=4 t=5 floats=7 ints=8 strings=9 type=20
  TensorProto  dims=1 data_type=2 float_data=4 int32_data=5 int64_data=7 name=8 raw_data=9 double_data=10
  ValueInfoProto name=1 type=2; TypeProto tensor_type=1; TypeProto.Tensor elem_type=1 shape=2
  TensorShapeProto dim=1; Dimension dim_value=1 dim_param=2
"""
import struct

import numpy as np

FLOAT, UINT8, INT8, UINT16, INT16, INT32, INT64, STRING, BOOL, FLOAT16, DOUBLE, UINT32, UINT64 = range(1, 14)
_NP = {FLOAT: np.float32, UINT8: np.uint8, INT8: np.int8, UINT16: np.uint16, INT16: np.int16, INT32: np.int32,
       INT64: np.int64, BOOL: np.bool_, FLOAT16: np.float16, DOUBLE: np.float64, UINT32: np.uint32, UINT64: np.uint64}
_ONNX = {np.dtype(v): k for k, v in _NP.items()}

ATTR_FLOAT, ATTR_INT, ATTR_STRING, ATTR_TENSOR, ATTR_FLOATS, ATTR_INTS, ATTR_STRINGS = 1, 2, 3, 4, 6, 7, 8


# ------------------------------------------------------------------------------------------ wire decoding
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _fields(buf):
    """Yields (field_number, wire_type, value) where value is int (varint / fixed) or a memoryview slice."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _packed_varints(v, wt):
    if wt == 0:
        return [_signed(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


def _packed_f32(v, wt):
    if wt == 5:
        return [struct.unpack("<f", v)[0]]
    return list(np.frombuffer(bytes(v), "<f4"))


class Tensor:
    __slots__ = ("name", "array")

    def __init__(self, name, array):
        self.name, self.array = name, array


def _parse_tensor(buf):
    dims, dtype, name, raw = [], FLOAT, "", None
    f32, i32, i64, f64 = [], [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v, wt)
        elif fno == 2:
            dtype = v
        elif fno == 4:
            f32 += _packed_f32(v, wt)
        elif fno == 5:
            i32 += _packed_varints(v, wt)
        elif fno == 7:
            i64 += _packed_varints(v, wt)
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 10:
            f64 += list(np.frombuffer(bytes(v), "<f8")) if wt == 2 else [struct.unpack("<d", v)[0]]
    if dtype not in _NP:
        raise ValueError(f"tensor {name}: unsupported ONNX data type {dtype}")
    npd = np.dtype(_NP[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, npd.newbyteorder("<")).astype(npd)
    elif dtype == FLOAT:
        arr = np.array(f32, np.float32)
    elif dtype == DOUBLE:
        arr = np.array(f64, np.float64)
    elif dtype == INT64:
        arr = np.array(i64, np.int64)
    elif dtype == FLOAT16:
        arr = np.array(i32, np.uint16).view(np.float16)
    else:
        arr = np.array(i32).astype(npd)
    return Tensor(name, arr.reshape(dims).copy())


def _parse_attr(buf):
    name, atype = "", 0
    f = i = s = t = None
    floats, ints, strings = [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            f = struct.unpack("<f", v)[0]
        elif fno == 3:
            i = _signed(v)
        elif fno == 4:
            s = bytes(v)
        elif fno == 5:
            t = _parse_tensor(v)
        elif fno == 7:
            floats += _packed_f32(v, wt)
        elif fno == 8:
            ints += _packed_varints(v, wt)
        elif fno == 9:
            strings.append(bytes(v))
        elif fno == 20:
            atype = v
    if atype == ATTR_FLOAT or (atype == 0 and f is not None):
        return name, float(f)
    if atype == ATTR_INT or (atype == 0 and i is not None):
        return name, int(i)
    if atype == ATTR_STRING or (atype == 0 and s is not None):
        return name, s.decode(errors="replace")
    if atype == ATTR_TENSOR or (atype == 0 and t is not None):
        return name, t.array
    if atype == ATTR_FLOATS or (atype == 0 and floats):
        return name, [float(x) for x in floats]
    if atype == ATTR_INTS or (atype == 0 and ints):
        return name, [int(x) for x in ints]
    if atype == ATTR_STRINGS:
        return name, [x.decode(errors="replace") for x in strings]
    return name, None


class Node:
    """Mirrors the NodeProto surface the reference touches: .name .op_type .input .output (+ .attrs dict)."""
    __slots__ = ("name", "op_type", "input", "output", "attrs", "domain")

    def __init__(self, op_type, inputs, outputs, name="", attrs=None, domain=""):
        self.op_type, self.input, self.output = op_type, list(inputs), list(outputs)
        self.name, self.attrs, self.domain = name, dict(attrs or {}), domain

    def __repr__(self):
        return f"Node({self.op_type}:{self.name} {self.input}->{self.output})"


def _parse_node(buf):
    n = Node("", [], [])
    for fno, wt, v in _fields(buf):
        if fno == 1:
            n.input.append(bytes(v).decode())
        elif fno == 2:
            n.output.append(bytes(v).decode())
        elif fno == 3:
            n.name = bytes(v).decode()
        elif fno == 4:
            n.op_type = bytes(v).decode()
        elif fno == 5:
            k, val = _parse_attr(v)
            n.attrs[k] = val
        elif fno == 7:
            n.domain = bytes(v).decode()
    return n


def _parse_value_info(buf):
    name, elem, shape = "", FLOAT, None
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 1:  # tensor_type
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            elem = v3
                        elif f3 == 2:
                            shape = []
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:
                                    dv = 0
                                    for f5, _, v5 in _fields(v4):
                                        if f5 == 1:
                                            dv = _signed(v5)
                                    shape.append(dv)
    return name, elem, shape


class Model:
    def __init__(self):
        self.ir_version, self.producer_name, self.opset = 8, "", {"": 13}
        self.graph_name = "graph"
        self.nodes, self.initializers = [], {}
        self.inputs, self.outputs, self.value_info = [], [], []  # lists of (name, elem_type, shape)


def load_model(path):
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    m = Model()
    m.opset = {}
    for fno, wt, v in _fields(buf):
        if fno == 1:
            m.ir_version = v
        elif fno == 2:
            m.producer_name = bytes(v).decode()
        elif fno == 8:
            dom, ver = "", 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = bytes(v2).decode()
                elif f2 == 2:
                    ver = v2
            m.opset[dom] = ver
        elif fno == 7:
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    m.nodes.append(_parse_node(v2))
                elif f2 == 2:
                    m.graph_name = bytes(v2).decode()
                elif f2 == 5:
                    t = _parse_tensor(v2)
                    m.initializers[t.name] = t.array
                elif f2 == 11:
                    m.inputs.append(_parse_value_info(v2))
                elif f2 == 12:
                    m.outputs.append(_parse_value_info(v2))
                elif f2 == 13:
                    m.value_info.append(_parse_value_info(v2))
    if not m.opset:
        m.opset = {"": 13}
    return m


def _spans(buf, pos, end):
    """Yields (field_number, wire_type, start, length) of the fields in buf[pos:end] without slicing (length-delimited fields:
    where their payload starts; other wire types: length 0)."""
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            _, pos = _varint(buf, pos)
            yield fno, wt, pos, 0
        elif wt == 1:
            pos += 8
            yield fno, wt, pos, 0
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            yield fno, wt, pos, ln
            pos += ln
        elif wt == 5:
            pos += 4
            yield fno, wt, pos, 0
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")


def scan_op_types(path):
    """{op_type: count} of a model file WITHOUT reading its tensors: the file is mapped, length-delimited fields other than the
    nodes are stepped over (a few milliseconds for ViT-B/16's 344 MB).  What a caller can know about a graph before it is loaded —
    the CLI decides from it whether the BLAS library's first call should start at once (__main__)."""
    import mmap
    counts = {}
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        for fno, wt, p0, n0 in _spans(mm, 0, len(mm)):
            if fno != 7 or wt != 2:
                continue
            for f2, w2, p1, n1 in _spans(mm, p0, p0 + n0):          # GraphProto
                if f2 != 1 or w2 != 2:
                    continue
                for f3, w3, p2, n2 in _spans(mm, p1, p1 + n1):      # NodeProto
                    if f3 == 4 and w3 == 2:
                        op = mm[p2:p2 + n2].decode()
                        counts[op] = counts.get(op, 0) + 1
    return counts


# ------------------------------------------------------------------------------------------ wire encoding
def _ev(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(fno, wt):
    return _ev((fno << 3) | wt)


def _len(fno, payload):
    return _key(fno, 2) + _ev(len(payload)) + payload


def _str(fno, s):
    return _len(fno, s.encode() if isinstance(s, str) else bytes(s))


def _int(fno, v):
    return _key(fno, 0) + _ev(int(v))


def _enc_tensor(name, arr):
    arr = np.asarray(arr)
    if arr.dtype not in _ONNX:
        raise ValueError(f"cannot encode dtype {arr.dtype}")
    out = b"".join(_int(1, d) for d in arr.shape)
    out += _int(2, _ONNX[arr.dtype]) + _str(8, name)
    out += _len(9, np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<")).tobytes())
    return out


def _enc_attr(name, val):
    out = _str(1, name)
    if isinstance(val, bool):
        val = int(val)
    if isinstance(val, float):
        out += _key(2, 5) + struct.pack("<f", val) + _int(20, ATTR_FLOAT)
    elif isinstance(val, (int, np.integer)):
        out += _int(3, val) + _int(20, ATTR_INT)
    elif isinstance(val, str):
        out += _str(4, val) + _int(20, ATTR_STRING)
    elif isinstance(val, np.ndarray):
        out += _len(5, _enc_tensor("", val)) + _int(20, ATTR_TENSOR)
    elif isinstance(val, (list, tuple)) and val and isinstance(val[0], float):
        out += b"".join(_key(7, 5) + struct.pack("<f", x) for x in val) + _int(20, ATTR_FLOATS)
    elif isinstance(val, (list, tuple)) and val and isinstance(val[0], str):
        out += b"".join(_str(9, x) for x in val) + _int(20, ATTR_STRINGS)
    elif isinstance(val, (list, tuple)):
        out += b"".join(_int(8, x) for x in val) + _int(20, ATTR_INTS)
    else:
        raise ValueError(f"attribute {name}: unsupported value {val!r}")
    return out


def _enc_value_info(name, elem, shape):
    t = _int(1, elem)
    if shape is not None:
        t += _len(2, b"".join(_len(1, _int(1, d)) for d in shape))
    return _str(1, name) + _len(2, _len(1, t))


def save_model(m, path):
    g = b""
    for n in m.nodes:
        nb = b"".join(_str(1, x) for x in n.input) + b"".join(_str(2, x) for x in n.output)
        nb += _str(3, n.name) + _str(4, n.op_type)
        nb += b"".join(_len(5, _enc_attr(k, v)) for k, v in n.attrs.items() if v is not None)
        if n.domain:
            nb += _str(7, n.domain)
        g += _len(1, nb)
    g += _str(2, m.graph_name)
    g += b"".join(_len(5, _enc_tensor(k, v)) for k, v in m.initializers.items())
    g += b"".join(_len(11, _enc_value_info(*vi)) for vi in m.inputs)
    g += b"".join(_len(12, _enc_value_info(*vi)) for vi in m.outputs)
    g += b"".join(_len(13, _enc_value_info(*vi)) for vi in m.value_info)
    out = _int(1, m.ir_version) + _str(2, m.producer_name or "dipoorlet_amd") + _len(7, g)
    out += b"".join(_len(8, _str(1, d) + _int(2, v)) for d, v in m.opset.items())
    with open(path, "wb") as f:
        f.write(out)
