"""Go encoding/gob in Python: an encoder that follows Go's rules (type-id allocation order,
type-definition messages, omitted zero fields) and a generic decoder.  Test infrastructure:
it fabricates reference-format sybil tables (tests/sybil_fixture.py) for the native loader
without a Go toolchain, and is itself pinned against the reference's golden gob files
(tests/test_gob.py).

Wire format: https://pkg.go.dev/encoding/gob "Encoding Details"; sybil's structs:
src/lib/column_store.go:22-74, table_column_info.go:13-24, table.go:10-23.
"""
import struct as _struct

BOOL, INT, UINT, FLOAT, BYTES, STRING, COMPLEX, INTERFACE = 1, 2, 3, 4, 5, 6, 7, 8
FIRST_USER_ID = 65


# ------------------------------------------------------------------ primitives
def enc_uint(u):
    if u < 128:
        return bytes([u])
    b = u.to_bytes((u.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def enc_int(i):
    u = (~i << 1) | 1 if i < 0 else i << 1
    return enc_uint(u & 0xFFFFFFFFFFFFFFFF)


def enc_float(f):
    bits = _struct.unpack("<Q", _struct.pack("<d", f))[0]
    rev = int.from_bytes(bits.to_bytes(8, "big")[::-1], "big")
    return enc_uint(rev)


def enc_string(s):
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    return enc_uint(len(b)) + b


# ------------------------------------------------------------------ type descriptions
class T:
    """A Go type as gob sees it."""
    id = 0


class Basic(T):
    def __init__(self, tid, name):
        self.id, self.name = tid, name


Bool, Int, Uint, Float, Bytes, String = (Basic(BOOL, "bool"), Basic(INT, "int"), Basic(UINT, "uint"),
                                         Basic(FLOAT, "float"), Basic(BYTES, "bytes"), Basic(STRING, "string"))


class Slice(T):
    def __init__(self, elem, name):
        self.elem, self.name = elem, name


class Map(T):
    def __init__(self, key, elem, name):
        self.key, self.elem, self.name = key, elem, name


class Struct(T):
    def __init__(self, name, fields):
        self.name, self.fields = name, fields  # fields: [(name, T)] exported fields in declaration order


def is_zero(t, v):
    if v is None:
        return True
    if isinstance(t, Basic):
        return v in (0, 0.0, False, "", b"")
    if isinstance(t, (Slice, Map)):
        return len(v) == 0
    return False  # structs are always sent (possibly empty)


class Encoder:
    """One gob.Encoder: allocates type ids like encoding/gob/type.go and emits definitions
    before first use like encoder.go:sendActualType."""

    def __init__(self):
        self.next_id = FIRST_USER_ID
        self.sent = set()
        self.out = bytearray()

    # type.go: a struct gets its id when created (before its fields are visited); slices and maps
    # get theirs in init(), after the element/key types have been built.
    def _assign(self, t):
        if isinstance(t, Basic) or t.id:
            return
        if isinstance(t, Struct):
            t.id = self.next_id
            self.next_id += 1
            for _, ft in t.fields:
                self._assign(ft)
        elif isinstance(t, Slice):
            self._assign(t.elem)
            t.id = self.next_id
            self.next_id += 1
        elif isinstance(t, Map):
            self._assign(t.key)
            self._assign(t.elem)
            t.id = self.next_id
            self.next_id += 1

    def _message(self, payload):
        self.out += enc_uint(len(payload)) + payload

    def _common(self, t):
        # CommonType{Name, Id}
        b = b""
        if t.name:
            b += enc_uint(1) + enc_string(t.name)
            b += enc_uint(1) + enc_int(t.id)
        else:
            b += enc_uint(2) + enc_int(t.id)
        return b + b"\x00"

    def _wiretype(self, t):
        if isinstance(t, Slice):     # wireType.SliceT (field 1): sliceType{CommonType, Elem}
            body = enc_uint(1) + self._common(t) + enc_uint(1) + enc_int(t.elem.id) + b"\x00"
            return enc_uint(2) + body + b"\x00"
        if isinstance(t, Struct):    # wireType.StructT (field 2): structType{CommonType, Field []fieldType}
            body = enc_uint(1) + self._common(t)
            if t.fields:
                body += enc_uint(1) + enc_uint(len(t.fields))
                for fname, ft in t.fields:
                    body += enc_uint(1) + enc_string(fname) + enc_uint(1) + enc_int(ft.id) + b"\x00"
            body += b"\x00"
            return enc_uint(3) + body + b"\x00"
        if isinstance(t, Map):       # wireType.MapT (field 3): mapType{CommonType, Key, Elem}
            body = enc_uint(1) + self._common(t) + enc_uint(1) + enc_int(t.key.id) + enc_uint(1) + enc_int(t.elem.id) + b"\x00"
            return enc_uint(4) + body + b"\x00"
        raise TypeError(t)

    def _send_type(self, t):
        if isinstance(t, Basic) or t.id in self.sent:
            return
        self.sent.add(t.id)
        self._message(enc_int(-t.id) + self._wiretype(t))
        if isinstance(t, Struct):
            for _, ft in t.fields:
                self._send_type(ft)
        elif isinstance(t, Slice):
            self._send_type(t.elem)
        elif isinstance(t, Map):
            self._send_type(t.key)
            self._send_type(t.elem)

    def _value(self, t, v):
        if isinstance(t, Basic):
            if t.id == BOOL:
                return enc_uint(1 if v else 0)
            if t.id == INT:
                return enc_int(int(v))
            if t.id == UINT:
                return enc_uint(int(v))
            if t.id == FLOAT:
                return enc_float(float(v))
            return enc_string(v)
        if isinstance(t, Slice):
            return enc_uint(len(v)) + b"".join(self._value(t.elem, x) for x in v)
        if isinstance(t, Map):
            items = v.items() if isinstance(v, dict) else v
            b = enc_uint(len(v))
            for k, x in items:
                b += self._value(t.key, k) + self._value(t.elem, x)
            return b
        if isinstance(t, Struct):
            b, prev = b"", -1
            for i, (fname, ft) in enumerate(t.fields):
                fv = v.get(fname) if isinstance(v, dict) else getattr(v, fname, None)
                if is_zero(ft, fv):
                    continue
                b += enc_uint(i - prev) + self._value(ft, fv if fv is not None else {})
                prev = i
            return b + b"\x00"
        raise TypeError(t)

    def encode(self, t, v):
        self._assign(t)
        self._send_type(t)
        payload = enc_int(t.id)
        if not isinstance(t, Struct):
            payload += b"\x00"
        self._message(payload + self._value(t, v))
        return bytes(self.out)


def encode(t, v):
    return Encoder().encode(t, v)


# ------------------------------------------------------------------ decoder (generic)
class Reader:
    def __init__(self, data, pos=0, end=None):
        self.d, self.p, self.end = data, pos, len(data) if end is None else end

    def uint(self):
        b = self.d[self.p]
        self.p += 1
        if b < 128:
            return b
        n = 256 - b
        v = int.from_bytes(self.d[self.p:self.p + n], "big")
        self.p += n
        return v

    def int(self):
        u = self.uint()
        return ~(u >> 1) if u & 1 else u >> 1

    def float(self):
        u = self.uint()
        return _struct.unpack("<d", u.to_bytes(8, "big"))[0]

    def bytes(self):
        n = self.uint()
        b = self.d[self.p:self.p + n]
        self.p += n
        return bytes(b)


def _read_common(r):
    name, f = "", -1
    while True:
        d = r.uint()
        if d == 0:
            return name
        f += d
        if f == 0:
            name = r.bytes().decode()
        else:
            r.int()


def _read_wiretype(r):
    td, f = None, -1
    while True:
        d = r.uint()
        if d == 0:
            return td
        f += d
        kind = ["array", "slice", "struct", "map", "opaque", "opaque", "opaque"][f]
        td = {"kind": kind, "name": "", "fields": []}
        g = -1
        while True:
            dd = r.uint()
            if dd == 0:
                break
            g += dd
            if g == 0:
                td["name"] = _read_common(r)
            elif kind == "struct" and g == 1:
                for _ in range(r.uint()):
                    fname, fid, h = "", 0, -1
                    while True:
                        d3 = r.uint()
                        if d3 == 0:
                            break
                        h += d3
                        if h == 0:
                            fname = r.bytes().decode()
                        else:
                            fid = r.int()
                    td["fields"].append((fname, fid))
            elif kind == "map" and g == 1:
                td["key"] = r.int()
            elif kind == "map" and g == 2:
                td["elem"] = r.int()
            elif g == 1:
                td["elem"] = r.int()
            elif g == 2:
                td["len"] = r.int()


def decode(data, want_types=False):
    """Decodes the first top-level value.  Structs -> dict (present fields only), maps -> dict."""
    types = {}
    order = []
    pos = 0

    def value(r, tid):
        if tid == BOOL:
            return r.uint() != 0
        if tid == INT:
            return r.int()
        if tid == UINT:
            return r.uint()
        if tid == FLOAT:
            return r.float()
        if tid == BYTES:
            return r.bytes()
        if tid == STRING:
            return r.bytes().decode("utf-8", "replace")
        if tid == INTERFACE:
            # name of the registered concrete type ("" = nil), its type id -- preceded by any type
            # definitions first needed here, each followed by a count to skip -- then the byte count of
            # the value and the value itself (encode.go:encodeInterface / decode.go:decodeInterface)
            name = r.bytes().decode()
            if not name:
                return None
            while True:
                cid = r.int()
                if cid >= 0:
                    break
                types[-cid] = _read_wiretype(r)
                order.append((-cid, types[-cid]["kind"], types[-cid]["name"]))
                r.uint()
            n = r.uint()
            end = r.p + n
            if cid not in types or types[cid]["kind"] != "struct":
                assert r.uint() == 0
            v = value(r, cid)
            assert r.p == end, (r.p, end)
            return {"@type": name, "@id": cid, "value": v}
        td = types[tid]
        if td["kind"] == "struct":
            out, f = {}, -1
            while True:
                d = r.uint()
                if d == 0:
                    return out
                f += d
                fname, fid = td["fields"][f]
                out[fname] = value(r, fid)
        if td["kind"] in ("slice", "array"):
            return [value(r, td["elem"]) for _ in range(r.uint())]
        if td["kind"] == "map":
            out = {}
            for _ in range(r.uint()):
                k = value(r, td["key"])
                out[k] = value(r, td["elem"])
            return out
        return r.bytes()

    while pos < len(data):
        hdr = Reader(data, pos)
        n = hdr.uint()
        r = Reader(data, hdr.p, hdr.p + n)
        pos = hdr.p + n
        tid = r.int()
        if tid < 0:
            types[-tid] = _read_wiretype(r)
            order.append((-tid, types[-tid]["kind"], types[-tid]["name"]))
            continue
        if tid not in types or types[tid]["kind"] != "struct":
            assert r.uint() == 0
        v = value(r, tid)
        return (v, order) if want_types else v
    raise ValueError("no value in stream")


# ------------------------------------------------------------------ sybil's on-disk structs
def _u32s():
    return Slice(Uint, "[]uint32")


def saved_int_column():
    bucket = Struct("SavedIntBucket", [("Value", Int), ("Records", _u32s())])
    return Struct("SavedIntColumn", [("Name", String), ("DeltaEncodedIDs", Bool), ("ValueEncoded", Bool),
                                     ("BucketEncoded", Bool), ("Bins", Slice(bucket, "[]sybil.SavedIntBucket")),
                                     ("Values", Slice(Int, "[]int64")), ("VERSION", Int)])


def saved_str_column():
    bucket = Struct("SavedStrBucket", [("Value", Int), ("Records", _u32s())])
    return Struct("SavedStrColumn", [("Name", String), ("DeltaEncodedIDs", Bool), ("BucketEncoded", Bool),
                                     ("Bins", Slice(bucket, "[]sybil.SavedStrBucket")), ("Values", Slice(Int, "[]int32")),
                                     ("StringTable", Slice(String, "[]string")), ("VERSION", Int)])


def saved_set_column():
    bucket = Struct("SavedSetBucket", [("Value", Int), ("Records", _u32s())])
    return Struct("SavedSetColumn", [("Name", String), ("Bins", Slice(bucket, "[]sybil.SavedSetBucket")),
                                     ("Values", Slice(Slice(Int, ""), "[][]int32")),
                                     ("StringTable", Slice(String, "[]string")), ("DeltaEncodedIDs", Bool),
                                     ("BucketEncoded", Bool), ("VERSION", Int)])


def _int_info():
    return Struct("", [("Min", Int), ("Max", Int), ("Avg", Float), ("M2", Float), ("Count", Int)])


def _str_info():
    return Struct("", [("TopStringCount", Map(Int, Int, "map[int32]int")), ("Cardinality", Int)])


def saved_column_info():
    return Struct("SavedColumnInfo", [("NumRecords", Int),
                                      ("StrInfoMap", Map(String, _str_info(), "SavedStrInfo")),
                                      ("IntInfoMap", Map(String, _int_info(), "SavedIntInfo"))])


def table_info():
    """The exported fields of sybil.Table that getSaveTable fills (table_io.go:72-78); the other
    exported fields are nil/zero and gob omits them, but their types are still described by Go --
    which a reader must tolerate and which this writer does not need to reproduce."""
    return Struct("Table", [("Name", String), ("KeyTable", Map(String, Int, "map[string]int16")),
                            ("KeyTypes", Map(Int, Int, "map[int16]int8")),
                            ("StrInfo", Map(Int, _str_info(), "StrInfoTable")),
                            ("IntInfo", Map(Int, _int_info(), "IntInfoTable"))])
