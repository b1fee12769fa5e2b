"""Reader / writer of the reference-dump format of tests/golden/ref_dump/README.md (test infrastructure).

A dump is one directory:
    frame.pgm         binary PGM (P5, maxval 255): the gray image handed to ORBextractor::operator()
    params.json       {"n_features", "scale_factor", "n_levels", "ini_th_fast", "min_th_fast",
                       "lapping_x0", "lapping_x1", "source"}
    keypoints.csv     header "x,y,octave,angle,response,size"; one cv::KeyPoint per line in OUTPUT ORDER;
                      floats printed with %.9g (round-trips an IEEE single)
    descriptors.bin   n x 32 bytes, row i = descriptor of keypoint i (cv::Mat CV_8U n x 32, row-major)
"""
import glob
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_dump")
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")])


class Dump:
    def __init__(self, image, params, kps, desc):
        self.image, self.params, self.kps, self.desc = image, params, kps, desc

    def oracle_params(self):
        keys = ("n_features", "scale_factor", "n_levels", "ini_th_fast", "min_th_fast", "lapping_x0", "lapping_x1")
        return {k: self.params[k] for k in keys if k in self.params}


def committed():
    """directories under tests/golden/ref_dump/ that hold a dump"""
    return sorted(os.path.dirname(p) for p in glob.glob(os.path.join(HERE, "*", "params.json")))


def read_pgm(path):
    data = open(path, "rb").read()
    if data[:2] != b"P5":
        raise ValueError(f"{path}: not a binary PGM")
    fields, pos = [], 2
    while len(fields) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        fields.append(int(data[pos:end]))
        pos = end
    w, h, maxval = fields
    if maxval != 255:
        raise ValueError(f"{path}: maxval {maxval} != 255")
    pix = np.frombuffer(data, np.uint8, count=w * h, offset=pos + 1)
    return pix.reshape(h, w).copy()


def write(directory, image, kps, desc, **params):
    os.makedirs(directory, exist_ok=True)
    h, w = image.shape
    with open(os.path.join(directory, "frame.pgm"), "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(image, np.uint8).tobytes())
    params.setdefault("lapping_x0", 0)
    params.setdefault("lapping_x1", 1000)
    json.dump(params, open(os.path.join(directory, "params.json"), "w"), indent=1)
    with open(os.path.join(directory, "keypoints.csv"), "w") as f:
        f.write("x,y,octave,angle,response,size\n")
        for k in kps:
            f.write("%.9g,%.9g,%d,%.9g,%.9g,%.9g\n" % (k["x"], k["y"], k["octave"], k["angle"], k["response"], k["size"]))
    np.ascontiguousarray(desc, np.uint8).tofile(os.path.join(directory, "descriptors.bin"))


def load(directory):
    image = read_pgm(os.path.join(directory, "frame.pgm"))
    params = json.load(open(os.path.join(directory, "params.json")))
    rows = [ln.strip().split(",") for ln in open(os.path.join(directory, "keypoints.csv")).read().splitlines()[1:] if ln.strip()]
    kps = np.zeros(len(rows), KP_DTYPE)
    for i, r in enumerate(rows):
        kps[i] = (np.float32(r[0]), np.float32(r[1]), np.float32(r[5]), np.float32(r[3]), np.float32(r[4]), int(r[2]))
    desc = np.fromfile(os.path.join(directory, "descriptors.bin"), np.uint8)
    if desc.size != 32 * len(rows):
        raise ValueError(f"{directory}: descriptors.bin holds {desc.size} bytes for {len(rows)} keypoints")
    return Dump(image, params, kps, desc.reshape(-1, 32))


def compare(dump, kps, desc):
    """-> list of human-readable differences (empty = bit-exact: every float field compared by its bits)"""
    out = []
    if len(kps) != len(dump.kps):
        out.append(f"keypoint count {len(kps)} != dump {len(dump.kps)}")
    n = min(len(kps), len(dump.kps))
    for name in ("x", "y", "octave", "angle", "response", "size"):
        a, b = np.ascontiguousarray(kps[name][:n]), np.ascontiguousarray(dump.kps[name][:n])
        bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0] if a.dtype.itemsize == 4 else np.nonzero(a != b)[0]
        for i in bad[:5]:
            out.append(f"keypoint {i}: {name} {a[i]!r} != dump {b[i]!r}")
    d = np.asarray(desc)[:n].reshape(-1, 32)
    bad = np.nonzero((d != dump.desc[:n]).any(axis=1))[0]
    for i in bad[:5]:
        out.append(f"descriptor {i}: {int(np.unpackbits(d[i] ^ dump.desc[i]).sum())} bits differ")
    if len(bad) > 5:
        out.append(f"... {len(bad)} descriptors differ in all")
    return out
