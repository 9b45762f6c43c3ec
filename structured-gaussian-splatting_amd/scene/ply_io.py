"""The reference's Gaussian PLY wire format (scene/gaussian_model.py:187-266) without
`plyfile`: binary little-endian, one `vertex` element, all properties float32 in the order
x y z nx ny nz f_dc_0..2 f_rest_0..(3(M-1)-1) opacity scale_0..2 rot_0..3, with f_rest stored CHANNEL-major
(the [P, M-1, 3] tensor is transposed to [P, 3, M-1] before flattening)."""
import numpy as np


def property_names(n_rest: int):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    return names


def write_gaussian_ply(path, xyz, f_dc, f_rest, opacity, scaling, rotation):
    """xyz[P,3] f_dc[P,1,3] f_rest[P,M-1,3] opacity[P,1] scaling[P,3] rotation[P,4] (numpy, raw parameters)."""
    P = xyz.shape[0]
    rest = np.ascontiguousarray(np.transpose(f_rest, (0, 2, 1))).reshape(P, -1)
    dc = np.ascontiguousarray(np.transpose(f_dc, (0, 2, 1))).reshape(P, -1)
    names = property_names(rest.shape[1])
    table = np.concatenate([xyz, np.zeros_like(xyz), dc, rest, opacity.reshape(P, 1), scaling, rotation], axis=1).astype("<f4")
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(np.ascontiguousarray(table).tobytes())


def read_gaussian_ply(path):
    """-> dict of numpy arrays with the shapes write_gaussian_ply takes, plus `names`."""
    with open(path, "rb") as fh:
        assert fh.readline().strip() == b"ply"
        fmt, count, names = None, None, []
        while True:
            line = fh.readline().decode("ascii").strip()
            if line == "end_header":
                break
            tok = line.split()
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                assert tok[1] == "vertex"
                count = int(tok[2])
            elif tok[0] == "property":
                assert tok[1] in ("float", "float32"), "Gaussian PLY files hold float32 properties only"
                names.append(tok[2])
        assert fmt == "binary_little_endian", fmt
        data = np.frombuffer(fh.read(count * len(names) * 4), dtype="<f4").reshape(count, len(names))
    col = {n: i for i, n in enumerate(names)}
    pick = lambda prefix: [col[n] for n in sorted((n for n in names if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))]
    rest_cols = pick("f_rest_")
    n_rest = len(rest_cols) // 3
    return dict(names=names, xyz=data[:, [col["x"], col["y"], col["z"]]].copy(),
                f_dc=data[:, pick("f_dc_")].reshape(count, 3, 1).transpose(0, 2, 1).copy(),
                f_rest=data[:, rest_cols].reshape(count, 3, n_rest).transpose(0, 2, 1).copy(),
                opacity=data[:, [col["opacity"]]].copy(), scaling=data[:, pick("scale_")].copy(),
                rotation=data[:, pick("rot_")].copy())
