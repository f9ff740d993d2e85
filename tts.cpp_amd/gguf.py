"""Self-contained GGUF v3 reader/writer (numpy only).

The reference reads its models through ggml's gguf_* API (src/models/loaders.cpp:52-76), which is
not vendored here; the `gguf` Python package is not installed either.  This module writes the same
container (layout per SURVEY.md A.3) so tests and bench can mint Parler-shaped model files with the
exact tensor names / KV keys the reference's loader looks for, and reads them back.
The C++ loader in host/gguf.cpp parses the same format; tests check the two against each other.
"""
import struct

import numpy as np

GGUF_MAGIC = b"GGUF"
GGUF_VERSION = 3
DEFAULT_ALIGNMENT = 32

# ggml tensor types (examples/quantize/README.md:48-55)
F32, F16, Q4_0, Q5_0, Q8_0 = 0, 1, 2, 6, 8
TYPE_NAMES = {F32: "F32", F16: "F16", Q4_0: "Q4_0", Q5_0: "Q5_0", Q8_0: "Q8_0"}
BLOCK = {F32: (1, 4), F16: (1, 2), Q4_0: (32, 18), Q5_0: (32, 22), Q8_0: (32, 34)}

# KV value types
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)
_SCALAR_FMT = {T_U8: "<B", T_I8: "<b", T_U16: "<H", T_I16: "<h", T_U32: "<I", T_I32: "<i", T_F32: "<f",
               T_BOOL: "<?", T_U64: "<Q", T_I64: "<q", T_F64: "<d"}


def nbytes(ttype, ne):
    n = int(np.prod(ne))
    be, bb = BLOCK[ttype]
    assert ne[0] % be == 0, f"ne[0]={ne[0]} not a multiple of block {be}"
    return n // be * bb


class Tensor:
    """A GGUF tensor: name, ggml type, ne (ne[0] fastest) and raw little-endian bytes."""

    def __init__(self, name, ttype, ne, data):
        self.name, self.type, self.ne = name, int(ttype), [int(x) for x in ne]
        self.data = data  # bytes / uint8 array
        assert len(self.raw()) == nbytes(self.type, self.ne), (name, len(self.raw()), nbytes(self.type, self.ne))

    def raw(self):
        return self.data if isinstance(self.data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(self.data).view(np.uint8).reshape(-1)

    @staticmethod
    def from_array(name, arr, ttype=F32):
        """arr is in PyTorch dimension order (slowest first); GGUF ne is the reverse."""
        arr = np.ascontiguousarray(arr)
        ne = list(reversed(arr.shape)) if arr.ndim else [1]
        if ttype == F32:
            return Tensor(name, F32, ne, arr.astype("<f4").tobytes())
        if ttype == F16:
            return Tensor(name, F16, ne, arr.astype("<f2").tobytes())
        raise ValueError("quantised tensors are built with Tensor(name, type, ne, blocks)")

    def to_f32(self):
        """fp32 view in PyTorch order (only for F32/F16)."""
        shape = list(reversed(self.ne))
        if self.type == F32:
            return np.frombuffer(bytes(self.raw()), dtype="<f4").reshape(shape)
        if self.type == F16:
            return np.frombuffer(bytes(self.raw()), dtype="<f2").astype(np.float32).reshape(shape)
        raise ValueError("dequantise with the oracle")


def _w_str(buf, s):
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    buf += struct.pack("<Q", len(b)) + b


def _w_value(buf, vtype, v):
    if vtype == T_STR:
        _w_str(buf, v)
    elif vtype == T_ARR:
        etype, items = v
        buf += struct.pack("<IQ", etype, len(items))
        if etype == T_STR:
            for it in items:
                _w_str(buf, it)
        else:
            buf += np.asarray(items).astype(np.dtype(_SCALAR_FMT[etype][1:]).newbyteorder("<")).tobytes()
    else:
        buf += struct.pack(_SCALAR_FMT[vtype], v)


def write(path, kv, tensors, alignment=DEFAULT_ALIGNMENT):
    """kv: list of (key, vtype, value); arrays are (T_ARR, (elem_type, items)). tensors: list[Tensor]."""
    buf = bytearray()
    buf += GGUF_MAGIC + struct.pack("<IQQ", GGUF_VERSION, len(tensors), len(kv))
    for key, vtype, v in kv:
        _w_str(buf, key)
        buf += struct.pack("<I", vtype)
        _w_value(buf, vtype, v)
    off = 0
    offsets = []
    for t in tensors:
        _w_str(buf, t.name)
        buf += struct.pack("<I", len(t.ne))
        for d in t.ne:
            buf += struct.pack("<Q", d)
        buf += struct.pack("<IQ", t.type, off)
        offsets.append(off)
        off += (len(t.raw()) + alignment - 1) // alignment * alignment
    pad = (-len(buf)) % alignment
    buf += b"\0" * pad
    with open(path, "wb") as f:
        f.write(buf)
        for t in tensors:
            raw = bytes(t.raw())
            f.write(raw)
            f.write(b"\0" * ((-len(raw)) % alignment))


class Reader:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        self.p = 0
        assert self._take(4) == GGUF_MAGIC, "not a GGUF file"
        self.version, n_tensors, n_kv = struct.unpack("<IQQ", self._take(20))
        assert self.version in (2, 3)
        self.kv = {}
        self.kv_types = {}
        for _ in range(n_kv):
            key = self._str()
            (vtype,) = struct.unpack("<I", self._take(4))
            self.kv[key] = self._value(vtype)
            self.kv_types[key] = vtype
        infos = []
        for _ in range(n_tensors):
            name = self._str()
            (nd,) = struct.unpack("<I", self._take(4))
            ne = list(struct.unpack("<%dQ" % nd, self._take(8 * nd)))
            ttype, off = struct.unpack("<IQ", self._take(12))
            infos.append((name, ttype, ne, off))
        align = int(self.kv.get("general.alignment", DEFAULT_ALIGNMENT))
        self.data_offset = (self.p + align - 1) // align * align
        self.tensors = {}
        self.order = []
        for name, ttype, ne, off in infos:
            n = nbytes(ttype, ne)
            start = self.data_offset + off
            self.tensors[name] = Tensor(name, ttype, ne, self.buf[start:start + n])
            self.order.append(name)

    def _take(self, n):
        b = self.buf[self.p:self.p + n]
        self.p += n
        return b

    def _str(self):
        (n,) = struct.unpack("<Q", self._take(8))
        return self._take(n).decode("utf-8", errors="surrogateescape")

    def _value(self, vtype):
        if vtype == T_STR:
            return self._str()
        if vtype == T_ARR:
            etype, n = struct.unpack("<IQ", self._take(12))
            if etype == T_STR:
                return [self._str() for _ in range(n)]
            dt = np.dtype(_SCALAR_FMT[etype][1:]).newbyteorder("<")
            return np.frombuffer(self._take(n * dt.itemsize), dtype=dt).copy()
        fmt = _SCALAR_FMT[vtype]
        return struct.unpack(fmt, self._take(struct.calcsize(fmt)))[0]
