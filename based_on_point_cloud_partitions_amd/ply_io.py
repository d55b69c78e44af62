"""3DGS PLY reader / writer for the merge step's exchange block (the data format either side of the path).

Byte-compatible with what the reference writes through plyfile in GaussianModel.save_ply
(scene/gaussian_model.py:189-220: binary little-endian, one `vertex` element, float32 properties in the order of
construct_list_of_attributes :189-201 -- x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3) and reads
files the way GaussianModel.load_ply does (:236-278: properties looked up by NAME, f_rest_/scale_/rot sorted by their
numeric suffix, normals ignored).  No plyfile dependency.  The in-memory layout is the [N, 59] block of merge.py
(PLY column order without the normals).
"""
import numpy as np

from .merge import ROW


def attribute_names(n_rest=45):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    return names


def save_ply(path, block):
    """block: [N, 59] float32 (numpy or torch) -> 3DGS .ply (normals written as zeros, gaussian_model.py:207)."""
    b = np.ascontiguousarray(block.detach().cpu().numpy() if hasattr(block, "detach") else block, dtype=np.float32)
    assert b.ndim == 2 and b.shape[1] == ROW
    n = b.shape[0]
    names = attribute_names()
    data = np.zeros((n, len(names)), dtype="<f4")
    data[:, 0:3] = b[:, 0:3]
    data[:, 6:] = b[:, 3:]
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {n}\n" + "".join(f"property float {a}\n" for a in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())


_NP = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
       "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def load_ply(path, max_sh_degree=3):
    """-> [N, 59] float32 numpy block (xyz, f_dc, f_rest, opacity, scale, rot)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, props, n, in_vertex = None, [], None, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties are not part of the 3DGS vertex element")
                props.append((tok[2], _NP[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian" or n is None:
            raise ValueError(f"unsupported PLY format {fmt}")
        rec = np.frombuffer(f.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
    num = lambda s: int(s.split("_")[-1])
    rest = sorted([p for p, _ in props if p.startswith("f_rest_")], key=num)
    assert len(rest) == 3 * (max_sh_degree + 1) ** 2 - 3  # gaussian_model.py:251
    scale = sorted([p for p, _ in props if p.startswith("scale_")], key=num)
    rot = sorted([p for p, _ in props if p.startswith("rot")], key=num)
    cols = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + rest + ["opacity"] + scale + rot
    return np.stack([rec[c].astype(np.float32) for c in cols], axis=1)
