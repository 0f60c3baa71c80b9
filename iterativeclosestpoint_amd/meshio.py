"""Mesh readers for the registration driver: the reader factory of the reference's Python
utilities (Wrapping/Python/icpmodules/util/io.py:16-37) without VTK.  Every reader yields a
vtkPolyData mirror holding float32 points and triangles (polygons are fanned from their first
corner, as vtkTriangleFilter does for convex polygons; vertices and lines are dropped, as
ICP/vtkImplicitPolyDataDistance2.cxx:41-48 drops them before the search).

Formats: .ply (ascii, binary little/big endian), .stl (ascii, binary; coincident corners
merged as vtkSTLReader does), .obj (v / f records, negative indices), .vtk (legacy POLYDATA,
ascii or binary), .vtp (XML PolyData: ascii, inline base64 and appended data, raw or zlib),
.g (BYU).  Host-side file parsing only; nothing here runs on the search path."""
import base64
import os
import re
import struct
import zlib

import numpy as np

from .vtkicp import vtkPolyData


class MeshFormatError(ValueError):
    pass


def _fan(polys):
    """Lists of corner indices -> (m, 3) int32 triangles (fan from the first corner)."""
    out = []
    for p in polys:
        for k in range(1, len(p) - 1):
            out.append((p[0], p[k], p[k + 1]))
    return np.asarray(out, np.int32).reshape(-1, 3)


def _fan_csr(conn, offsets):
    """Connectivity + end offsets (VTK's XML layout) -> triangles."""
    conn = np.asarray(conn, np.int64)
    ends = np.asarray(offsets, np.int64)
    starts = np.concatenate([[0], ends[:-1]])
    sizes = ends - starts
    if len(sizes) and np.all(sizes == 3):
        return conn.reshape(-1, 3).astype(np.int32)
    return _fan([conn[s:e] for s, e in zip(starts, ends)])


def _finish(points, tris):
    points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, np.int32).reshape(-1, 3)
    if len(tris) and (tris.min() < 0 or tris.max() >= len(points)):
        raise MeshFormatError("a polygon refers to a point that is not in the file")
    pd = vtkPolyData()
    pd.SetPoints(points)
    pd.SetPolys(tris)
    return pd


# ---------------------------------------------------------------------------------- PLY
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2",
              "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8",
              "float64": "f8"}


def read_ply(path):
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise MeshFormatError("%s: not a PLY file" % path)
    body = data[data.index(b"\n", end) + 1:]
    fmt, elements = None, []
    for line in data[:end].decode("ascii", "replace").splitlines():
        w = line.split()
        if not w:
            continue
        if w[0] == "format":
            fmt = w[1]
        elif w[0] == "element":
            elements.append({"name": w[1], "count": int(w[2]), "props": []})
        elif w[0] == "property" and elements:
            if w[1] == "list":
                elements[-1]["props"].append(("list", w[2], w[3], w[4]))
            else:
                elements[-1]["props"].append(("scalar", w[1], w[2]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise MeshFormatError("%s: unknown PLY format %r" % (path, fmt))
    points, polys = None, []
    if fmt == "ascii":
        tok = body.split()
        pos = 0
        for el in elements:
            rows = []
            for _ in range(el["count"]):
                row = {}
                for pr in el["props"]:
                    if pr[0] == "scalar":
                        row[pr[2]] = float(tok[pos])
                        pos += 1
                    else:
                        n = int(tok[pos])
                        row[pr[3]] = [int(t) for t in tok[pos + 1:pos + 1 + n]]
                        pos += 1 + n
                rows.append(row)
            if el["name"] == "vertex":
                points = np.array([[r["x"], r["y"], r["z"]] for r in rows], np.float64)
            elif el["name"] == "face":
                key = "vertex_indices" if rows and "vertex_indices" in rows[0] else "vertex_index"
                polys = [r[key] for r in rows]
    else:
        bo = "<" if fmt == "binary_little_endian" else ">"
        pos = 0
        for el in elements:
            scalar_only = all(pr[0] == "scalar" for pr in el["props"])
            if scalar_only:
                dt = np.dtype([(pr[2], bo + _PLY_TYPES[pr[1]]) for pr in el["props"]])
                arr = np.frombuffer(body, dt, el["count"], pos)
                pos += dt.itemsize * el["count"]
                if el["name"] == "vertex":
                    points = np.stack([arr["x"], arr["y"], arr["z"]], -1).astype(np.float64)
                continue
            for _ in range(el["count"]):
                for pr in el["props"]:
                    if pr[0] == "scalar":
                        pos += np.dtype(_PLY_TYPES[pr[1]]).itemsize
                    else:
                        ct, it = np.dtype(bo + _PLY_TYPES[pr[1]]), np.dtype(bo + _PLY_TYPES[pr[2]])
                        n = int(np.frombuffer(body, ct, 1, pos)[0])
                        pos += ct.itemsize
                        idx = np.frombuffer(body, it, n, pos)
                        pos += it.itemsize * n
                        if el["name"] == "face" and pr[3] in ("vertex_indices", "vertex_index"):
                            polys.append(idx.astype(np.int64))
    if points is None:
        raise MeshFormatError("%s: no vertex element" % path)
    return _finish(points, _fan(polys))


# ---------------------------------------------------------------------------------- STL
def _merge_corners(corners):
    """(3m, 3) corner coordinates -> unique points (first-seen order) and triangles."""
    corners = np.ascontiguousarray(corners, np.float32).reshape(-1, 3)
    uniq, first, inv = np.unique(corners, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)            # unique rows in order of first appearance
    rank = np.empty(len(order), np.int64)
    rank[order] = np.arange(len(order))
    return uniq[order], rank[inv.reshape(-1)].reshape(-1, 3)


def read_stl(path):
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) >= 84:
        n = struct.unpack_from("<I", data, 80)[0]
        if 84 + 50 * n == len(data):     # binary: 80-byte header, count, 50-byte records
            rec = np.frombuffer(data, np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                                n, 84)
            pts, tris = _merge_corners(rec["v"].reshape(-1, 3))
            return _finish(pts, tris)
    text = data.decode("ascii", "replace")
    v = re.findall(r"vertex\s+(\S+)\s+(\S+)\s+(\S+)", text)
    if not v or len(v) % 3:
        raise MeshFormatError("%s: not an STL file" % path)
    pts, tris = _merge_corners(np.array(v, np.float64))
    return _finish(pts, tris)


# ---------------------------------------------------------------------------------- OBJ
def read_obj(path):
    pts, polys = [], []
    with open(path, "r", errors="replace") as fh:
        for line in fh:
            w = line.split()
            if not w:
                continue
            if w[0] == "v":
                pts.append([float(t) for t in w[1:4]])
            elif w[0] == "f":
                idx = []
                for t in w[1:]:
                    i = int(t.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(pts) + i)
                polys.append(idx)
    if not pts:
        raise MeshFormatError("%s: no vertices" % path)
    return _finish(np.array(pts, np.float64), _fan(polys))


# ---------------------------------------------------------------------------------- BYU
def read_byu(path):
    with open(path, "r", errors="replace") as fh:
        tok = fh.read().split()
    nparts, npts, npolys, nedges = (int(t) for t in tok[:4])
    pos = 4 + 2 * nparts
    pts = np.array(tok[pos:pos + 3 * npts], np.float64).reshape(-1, 3)
    pos += 3 * npts
    polys, cur = [], []
    for t in tok[pos:pos + nedges]:
        i = int(t)
        cur.append(abs(i) - 1)
        if i < 0:                        # a negative index closes the polygon
            polys.append(cur)
            cur = []
    if len(polys) != npolys:
        raise MeshFormatError("%s: %d polygons announced, %d found" % (path, npolys, len(polys)))
    return _finish(pts, _fan(polys))


# ------------------------------------------------------------------------- legacy .vtk
_VTK_TYPES = {"float": "f4", "double": "f8", "int": "i4", "unsigned_int": "u4", "long": "i8",
              "vtktypeint64": "i8", "vtktypeint32": "i4", "short": "i2", "unsigned_short": "u2",
              "char": "i1", "unsigned_char": "u1", "unsigned_long": "u8"}


def read_vtk(path):
    with open(path, "rb") as fh:
        data = fh.read()
    lines = data.split(b"\n", 4)
    if len(lines) < 5 or not lines[0].startswith(b"# vtk DataFile"):
        raise MeshFormatError("%s: not a legacy VTK file" % path)
    binary = lines[2].strip().upper() == b"BINARY"
    if b"POLYDATA" not in lines[3].upper():
        raise MeshFormatError("%s: not a POLYDATA file" % path)
    pos = len(data) - len(lines[4])

    def token():
        nonlocal pos
        while pos < len(data) and data[pos:pos + 1].isspace():
            pos += 1
        s = pos
        while pos < len(data) and not data[pos:pos + 1].isspace():
            pos += 1
        return data[s:pos].decode("ascii", "replace")

    def values(n, dt):
        nonlocal pos
        if binary:
            pos += 1                     # the newline that ends the keyword line
            a = np.frombuffer(data, np.dtype(">" + dt), n, pos)
            pos += a.nbytes
            return a
        return np.array([token() for _ in range(n)], np.float64).astype(dt)

    pts, tris = None, np.zeros((0, 3), np.int32)
    while True:
        kw = token().upper()
        if not kw:
            break
        if kw == "POINTS":
            n, ty = int(token()), token().lower()
            pts = values(3 * n, _VTK_TYPES[ty]).reshape(-1, 3)
        elif kw == "POLYGONS":
            a, b = int(token()), int(token())
            save = pos
            nxt = token().upper()
            if nxt == "OFFSETS":         # the layout written since VTK 9: offsets, connectivity
                off = values(a, _VTK_TYPES[token().lower()])
                if token().upper() != "CONNECTIVITY":
                    raise MeshFormatError("%s: CONNECTIVITY expected" % path)
                conn = values(b, _VTK_TYPES[token().lower()])
                tris = _fan_csr(conn, off[1:])
            else:
                pos = save
                flat = values(b, "i4")
                polys, i = [], 0
                for _ in range(a):
                    k = int(flat[i])
                    polys.append(flat[i + 1:i + 1 + k].astype(np.int64))
                    i += 1 + k
                tris = _fan(polys)
        elif kw in ("POINT_DATA", "CELL_DATA"):
            break                        # attributes are not needed for registration
        elif kw in ("VERTICES", "LINES", "TRIANGLE_STRIPS"):
            a, b = int(token()), int(token())
            save = pos
            if token().upper() == "OFFSETS":
                values(a, _VTK_TYPES[token().lower()])
                token()
                values(b, _VTK_TYPES[token().lower()])
            else:
                pos = save
                values(b, "i4")
    if pts is None:
        raise MeshFormatError("%s: no POINTS" % path)
    return _finish(pts, tris)


# ---------------------------------------------------------------------------- XML .vtp
_XML_TYPES = {"Float32": "f4", "Float64": "f8", "Int8": "i1", "UInt8": "u1", "Int16": "i2",
              "UInt16": "u2", "Int32": "i4", "UInt32": "u4", "Int64": "i8", "UInt64": "u8"}


def _xml_blocks(raw, hdr, compressed, encoded):
    """One data array's bytes from its [header | payload] block (VTK XML binary layout)."""
    hsz = np.dtype(hdr).itemsize
    if not compressed:
        if encoded:
            n = int(np.frombuffer(base64.b64decode(raw[:((hsz + 2) // 3) * 4]), hdr, 1)[0])
            return base64.b64decode(raw[:((hsz + n + 2) // 3) * 4])[hsz:hsz + n]
        n = int(np.frombuffer(raw, hdr, 1)[0])
        return raw[hsz:hsz + n]
    if encoded:
        first = base64.b64decode(raw[:((3 * hsz + 2) // 3) * 4])
        nblocks = int(np.frombuffer(first, hdr, 1)[0])
        hlen = hsz * (3 + nblocks)
        henc = ((hlen + 2) // 3) * 4
        head = np.frombuffer(base64.b64decode(raw[:henc]), hdr, 3 + nblocks)
        body = base64.b64decode(raw[henc:henc + ((int(head[3:].sum()) + 2) // 3) * 4])
    else:
        nblocks = int(np.frombuffer(raw, hdr, 1)[0])
        head = np.frombuffer(raw, hdr, 3 + nblocks)
        body = raw[hsz * (3 + nblocks):]
    out, pos = [], 0
    for sz in head[3:]:
        out.append(zlib.decompress(body[pos:pos + int(sz)]))
        pos += int(sz)
    return b"".join(out)


def read_vtp(path):
    with open(path, "rb") as fh:
        data = fh.read()
    head = re.search(rb"<VTKFile([^>]*)>", data)
    if not head:
        raise MeshFormatError("%s: not a VTK XML file" % path)
    attrs = dict(re.findall(r'(\w+)\s*=\s*"([^"]*)"', head.group(1).decode()))
    bo = "<" if attrs.get("byte_order", "LittleEndian") == "LittleEndian" else ">"
    hdr = bo + _XML_TYPES[attrs.get("header_type", "UInt32")]
    compressed = "compressor" in attrs
    appended, app_encoded = None, True
    m = re.search(rb"<AppendedData([^>]*)>\s*_", data)
    if m:
        app_encoded = b"raw" not in m.group(1)
        appended = data[m.end():]
        if app_encoded:
            appended = appended[:appended.index(b"<")].strip()

    def arrays(section):
        sec = re.search(rb"<" + section + rb"\b[^>]*>(.*?)</" + section + rb">", data, re.S)
        out = {}
        if not sec:
            return out
        for am in re.finditer(rb"<DataArray([^>]*?)(/>|>(.*?)</DataArray>)", sec.group(1), re.S):
            a = dict(re.findall(r'(\w+)\s*=\s*"([^"]*)"', am.group(1).decode()))
            dt = np.dtype(bo + _XML_TYPES[a["type"]])
            fmt = a.get("format", "ascii")
            if fmt == "ascii":
                v = np.array((am.group(3) or b"").split(), np.float64).astype(dt)
            elif fmt == "binary":
                v = np.frombuffer(_xml_blocks((am.group(3) or b"").strip(), hdr, compressed, True), dt)
            else:
                off = int(a["offset"])
                v = np.frombuffer(_xml_blocks(appended[off:], hdr, compressed, app_encoded), dt)
            out[a.get("Name", "")] = v
        return out

    pts = arrays(b"Points")
    if not pts:
        raise MeshFormatError("%s: no Points" % path)
    polys = arrays(b"Polys")
    points = next(iter(pts.values())).reshape(-1, 3)
    tris = np.zeros((0, 3), np.int32)
    if "connectivity" in polys and len(polys["connectivity"]):
        tris = _fan_csr(polys["connectivity"], polys["offsets"])
    return _finish(points, tris)


# ------------------------------------------------------------------------------ factory
_READERS = {".ply": read_ply, ".vtp": read_vtp, ".stl": read_stl, ".vtk": read_vtk,
            ".g": read_byu, ".obj": read_obj}


class _Reader:
    """SetFileName / Update / GetOutput, like the VTK readers the factory returns."""

    def __init__(self, fn, path=None):
        self._fn, self._path, self._out = fn, path, None

    def SetFileName(self, path):
        self._path, self._out = path, None

    def GetFileName(self):
        return self._path

    def Update(self):
        if self._out is None:
            self._out = self._fn(self._path)

    def GetOutput(self):
        self.Update()
        return self._out


class vtkPolyDataReaderFactory:
    """Reader by file extension (Wrapping/Python/icpmodules/util/io.py:16-37)."""

    def __new__(cls):
        raise TypeError("Static classes cannot be instantiated")

    @staticmethod
    def CreateReader(inputFileName):
        ext = os.path.splitext(inputFileName)[1].lower()
        if ext not in _READERS:
            raise NameError("Filetype not supported")
        return _Reader(_READERS[ext], inputFileName)

    @staticmethod
    def SupportedExtensions():
        return [".ply", ".vtp", ".stl", ".vtk", ".g", ".obj"]

    @staticmethod
    def CreateReaders(inputFileNames, append=False):
        readers = [vtkPolyDataReaderFactory.CreateReader(f) for f in inputFileNames]
        if not append:
            return readers
        return _Reader(lambda _: append_polydata([r.GetOutput() for r in readers]), "")


def append_polydata(parts):
    """vtkAppendPolyData for points and triangles."""
    pts, tris, base = [], [], 0
    for pd in parts:
        pts.append(np.asarray(pd.GetPoints(), np.float32))
        tris.append(np.asarray(pd.GetPolys(), np.int32) + base)
        base += len(pts[-1])
    return _finish(np.concatenate(pts), np.concatenate(tris))


def read_mesh(path):
    """(points float32 (n,3), triangles int32 (m,3)) of any supported file."""
    pd = vtkPolyDataReaderFactory.CreateReader(path).GetOutput()
    return pd.GetPoints(), pd.GetPolys()
