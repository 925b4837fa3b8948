"""Minimal PLY vertex-element reader / writer (the reference uses the `plyfile` package, which is not a
dependency here).  Written files are what `PlyData([PlyElement.describe(elements, "vertex")]).write(path)` produces
for an all-float32 structured array: `format binary_little_endian 1.0`, one `element vertex N`, one
`property float <name>` per column (gm_background.py:208-225).  The reader accepts binary little-endian and ASCII
files with scalar properties of the usual PLY types and returns float64 columns by name."""
from __future__ import annotations

import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def write_vertex_ply(path, names, columns):
    a = np.ascontiguousarray(columns, dtype="<f4")
    if a.ndim != 2 or a.shape[1] != len(names):
        raise ValueError("columns must be [N, len(names)]")
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {a.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(a.tobytes())


def read_vertex_ply(path):
    """-> (names, {name: float64 array [N]}) of the first element, which must be `vertex`."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex, seen_element = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if seen_element and in_vertex:
                    in_vertex = False  # later elements (faces ...) are ignored; vertex data comes first
                elif not seen_element:
                    if tok[1] != "vertex":
                        raise ValueError(f"{path}: first element is {tok[1]}, expected vertex")
                    n, in_vertex, seen_element = int(tok[2]), True, True
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if fmt == "binary_little_endian":
            dt = np.dtype([(nm, "<" + t) for nm, t in props])
            data = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
            cols = {nm: data[nm].astype(np.float64) for nm in names}
        elif fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n, ndmin=2) if n else np.zeros((0, len(names)))
            cols = {nm: rows[:, i].astype(np.float64) for i, nm in enumerate(names)}
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    return names, cols
