"""3DGS ``.ply`` reader / writer without ``plyfile`` (absent in this image): the on-disk format either
side of the hot path.

Layout as the reference reads and writes it (``src/models/gaussian_model.py:98-139,155-185``): one ``vertex``
element, float32 properties ``x y z nx ny nz f_dc_0..2 f_rest_0..(3K-1) opacity scale_0..2 rot_0..3`` in
binary little endian.  ``f_rest`` is stored channel-major on disk (``(P, 3, K)``) and transposed to the
coefficient-major ``(P, K, 3)`` tensors the model keeps (``gaussian_model.py:115-116,131-134``);
``opacity`` is the raw logit, ``scale`` the log standard deviations, ``rot`` the (w, x, y, z) quaternion.
ASCII ply files are read too.
"""
from __future__ import annotations

import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def _read_header(f, path):
    """-> (format, vertex count, [(property, numpy type)], True when the vertex element is the FIRST element of the file)."""
    if f.readline().strip() != b"ply":
        raise ValueError(f"{path}: not a ply file")
    fmt, n, props, in_vertex, n_elements, vertex_first = None, 0, [], False, 0, False
    while True:
        line = f.readline()
        if not line:
            raise ValueError(f"{path}: truncated ply header")
        tok = line.decode("ascii", "replace").split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            n_elements += 1
            if in_vertex:
                n = int(tok[2])
                vertex_first = n_elements == 1
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise ValueError(f"{path}: list properties in the vertex element are not supported")
            props.append((tok[2], _PLY_TYPES[tok[1]]))
        elif tok[0] == "end_header":
            break
    return fmt, n, props, vertex_first


def read_ply_property_names(path):
    """Names of the vertex element's properties, from the header alone (the file's type, file_loader.check_point_cloud_type)."""
    with open(path, "rb") as f:
        return tuple(name for name, _ in _read_header(f, path)[2])


def read_ply_vertices(path) -> np.ndarray:
    """Return the ``vertex`` element as a structured numpy array."""
    with open(path, "rb") as f:
        fmt, n, props, _ = _read_header(f, path)
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n, ndmin=2)
            out = np.empty(n, dtype=[(name, "<" + t) for name, t in props])
            for k, (name, _) in enumerate(props):
                out[name] = data[:, k]
            return out
        endian = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(name, endian + t) for name, t in props])
        out = np.fromfile(f, dtype=dt, count=n)
        if out.shape[0] != n:
            raise ValueError(f"{path}: expected {n} vertices, found {out.shape[0]}")
        return out


def is_gaussian_ply(vertices: np.ndarray) -> bool:
    names = vertices.dtype.names or ()
    return all(k in names for k in ("x", "y", "z", "opacity", "f_dc_0", "scale_0", "rot_0"))


def load_gaussian_arrays(path) -> dict:
    """``GaussianModel.from_ply`` arithmetic on the host: the five level-0 arrays + scaling/rotation.

    xyz (P,3), colors (P,3) = SH DC, sh (P, 3K) coefficient-major flattened, opacity (P,) raw, cov6 (P,6) from
    ``R diag(exp(scale))^2 R^T`` (``gaussian_model.py:34-38,139``, ``general_utils.py:43-80``), scale (P,3) log, rot (P,4)."""
    v = read_ply_vertices(path)
    if not is_gaussian_ply(v):
        raise ValueError(f"{path}: not a Gaussian-splat ply (missing opacity / f_dc / scale / rot properties)")
    names = v.dtype.names
    P = v.shape[0]
    xyz = np.stack([v["x"], v["y"], v["z"]], 1).astype(np.float32)
    dc = np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], 1).astype(np.float32)
    rest_names = sorted([k for k in names if k.startswith("f_rest_")], key=lambda s: int(s.split("_")[-1]))
    K3 = len(rest_names)
    sh_degree = int(round(((K3 + 3) / 3) ** 0.5 - 1))
    K = (sh_degree + 1) ** 2 - 1
    rest = np.stack([v[k] for k in rest_names], 1).astype(np.float32) if K3 else np.zeros((P, 0), np.float32)
    rest = rest.reshape(P, 3, K).transpose(0, 2, 1).reshape(P, 3 * K)          # (P,3,K) on disk -> (P,K,3) flattened
    scale = np.stack([v[k] for k in sorted([k for k in names if k.startswith("scale_")], key=lambda s: int(s.split("_")[-1]))], 1).astype(np.float32)
    rot = np.stack([v[k] for k in sorted([k for k in names if k.startswith("rot")], key=lambda s: int(s.split("_")[-1]))], 1).astype(np.float32)
    q = rot / np.linalg.norm(rot, axis=1, keepdims=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((P, 3, 3), np.float32)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    L = R * np.exp(scale)[:, None, :]
    C = L @ L.transpose(0, 2, 1)
    cov6 = np.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], 1).astype(np.float32)
    return {"xyz": xyz, "color": dc, "sh": np.ascontiguousarray(rest), "opacity": np.asarray(v["opacity"], np.float32),
            "cov6": cov6, "scale": scale, "rot": rot, "sh_degree": sh_degree}


def load_gaussian_device(path, device=0, chunk_rows: int = 1 << 18, timing: dict | None = None) -> dict:
    """``GaussianModel.from_ply`` straight into device SoA (the reference reads a ``.ply`` onto ``cuda:0``, ``file_loader.py:53-66``,
    ``gaussian_model.py:98-139``): the file's vertex rows are read in chunks into two PINNED host buffers, copied to HBM with
    asynchronous copies, and scattered there by one kernel per chunk (``gsr_ply_unpack``: the five level-0 arrays + scale /
    rotation, the SH block transposed, the covariance built on the device) -- disk read, PCIe copy and the scatter of consecutive
    chunks overlap, and no full-size host copy of the cloud ever exists.  Returns the dict of ``load_gaussian_arrays`` with CUDA
    tensors; they can be handed to ``gsr_hem_set_level0`` in place (``on_device = 2``).  Binary little-endian files whose wanted
    properties are float32 (what 3DGS writes); anything else raises -- ``load_gaussian_arrays`` is the general host reader."""
    import time
    import ctypes as C
    import torch
    from .. import _lib
    L = _lib.load(require_device=True)
    t0 = time.perf_counter()
    with open(path, "rb") as f:
        fmt, n, props, vertex_first = _read_header(f, path)
        if fmt != "binary_little_endian" or not vertex_first:
            raise ValueError(f"{path}: the device loader reads binary little-endian files with the vertex element first")
        off, pos = {}, 0
        for name, t in props:
            off[name] = (pos, t)
            pos += np.dtype(t).itemsize
        row_bytes = pos
        need = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
        for k in need:
            if k not in off:
                raise ValueError(f"{path}: not a Gaussian-splat ply (missing property {k})")
            if off[k][1] != "f4":
                raise ValueError(f"{path}: property {k} is not float32")
        rest = sorted([k for k in off if k.startswith("f_rest_")], key=lambda q: int(q.split("_")[-1]))
        K3 = len(rest)
        sh_degree = int(round(((K3 + 3) / 3) ** 0.5 - 1))
        K = (sh_degree + 1) ** 2 - 1
        if 3 * K != K3 or any(off[k] != (off[rest[0]][0] + 4 * i, "f4") for i, k in enumerate(rest)):
            raise ValueError(f"{path}: f_rest_* must be {3 * K} consecutive float32 properties")
        offsets = (C.c_int32 * 15)(*([off[k][0] for k in need] + [off[rest[0]][0] if K3 else 0]))
        dev = torch.device("cuda", int(device))
        out = {"xyz": torch.empty((n, 3), dtype=torch.float32, device=dev), "color": torch.empty((n, 3), dtype=torch.float32, device=dev),
               "sh": torch.empty((n, 3 * K), dtype=torch.float32, device=dev), "opacity": torch.empty((n,), dtype=torch.float32, device=dev),
               "cov6": torch.empty((n, 6), dtype=torch.float32, device=dev), "scale": torch.empty((n, 3), dtype=torch.float32, device=dev),
               "rot": torch.empty((n, 4), dtype=torch.float32, device=dev), "sh_degree": sh_degree}
        rows = max(1, min(int(chunk_rows), n))
        pinned = [torch.empty(rows * row_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        staged = [torch.empty(rows * row_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        free = [torch.cuda.Event(), torch.cuda.Event()]          # pinned[b] may be overwritten once its copy has left
        stream = torch.cuda.current_stream(dev)
        done, b = 0, 0
        while done < n:
            m = min(rows, n - done)
            if done >= 2 * rows:
                free[b].synchronize()
            got = f.readinto(memoryview(pinned[b].numpy())[: m * row_bytes])
            if got != m * row_bytes:
                raise ValueError(f"{path}: expected {n} vertices, the file ends after {done + got // row_bytes}")
            staged[b][: m * row_bytes].copy_(pinned[b][: m * row_bytes], non_blocking=True)
            free[b].record(stream)
            o = lambda t, w: C.c_void_p(t.data_ptr() + 4 * w * done)
            _lib.check(L.gsr_ply_unpack(C.c_void_p(staged[b].data_ptr()), m, row_bytes, offsets, K, o(out["xyz"], 3), o(out["color"], 3),
                                        o(out["sh"], 3 * K) if K else None, o(out["opacity"], 1), o(out["scale"], 3), o(out["rot"], 4), o(out["cov6"], 6),
                                        int(device), C.c_void_p(stream.cuda_stream)), "gsr_ply_unpack")
            done += m
            b ^= 1
        stream.synchronize()
    if timing is not None:
        timing.update(seconds=time.perf_counter() - t0, bytes=n * row_bytes, splats=n)
    return out


def save_gaussian_ply(path, xyz, colors, sh, opacity, scale, rot):
    """``GaussianModel.save_ply`` (``gaussian_model.py:169-185``): binary little endian, normals zero."""
    xyz = np.asarray(xyz, np.float32)
    P = xyz.shape[0]
    sh = np.asarray(sh, np.float32).reshape(P, -1)
    K = sh.shape[1] // 3
    rest = sh.reshape(P, K, 3).transpose(0, 2, 1).reshape(P, 3 * K)            # back to channel-major on disk
    cols = [xyz, np.zeros((P, 3), np.float32), np.asarray(colors, np.float32).reshape(P, 3), rest,
            np.asarray(opacity, np.float32).reshape(P, 1), np.asarray(scale, np.float32).reshape(P, -1),
            np.asarray(rot, np.float32).reshape(P, -1)]
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(3 * K)] + ["opacity"]
    names += [f"scale_{i}" for i in range(cols[5].shape[1])] + [f"rot_{i}" for i in range(cols[6].shape[1])]
    data = np.ascontiguousarray(np.concatenate(cols, 1).astype("<f4"))
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(f"element vertex {P}\n".encode())
        for nme in names:
            f.write(f"property float {nme}\n".encode())
        f.write(b"end_header\n")
        data.tofile(f)


def save_input_ply(path, xyz, rgb, normals=None):
    """A sparse *input* cloud as COLMAP / the 3DGS pipeline store it (``points3D.ply``): float ``x y z`` (+ ``nx ny nz``),
    uchar ``red green blue``, binary little endian -- the file type ``file_loader.check_point_cloud_type`` calls INPUT."""
    xyz = np.asarray(xyz, np.float32)
    P = xyz.shape[0]
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    v = np.empty(P, dtype=fields)
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if normals is not None:
        nrm = np.asarray(normals, np.float32)
        v["nx"], v["ny"], v["nz"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
    rgb = np.asarray(rgb)
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(f"element vertex {P}\n".encode())
        for name, t in fields:
            f.write(f"property {'uchar' if t == 'u1' else 'float'} {name}\n".encode())
        f.write(b"end_header\n")
        v.tofile(f)
