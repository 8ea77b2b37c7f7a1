"""KITTI odometry on-disk formats as the reference reads and writes them, and a sequence runner on top.

  calib.txt          one line per camera, "P0: " + 12 numbers (row-major 3x4); the first two lines are the left and
                     right grey cameras (Source/VOClass.cpp:102-138, Source/VOClassUtils.cpp:6-18: the first word of
                     a line is skipped, whatever it is).
  poses/XX.txt       12 numbers per line, row-major 3x4 [R|t] of the left camera; the ground-truth trajectory is the
                     translation column (Source/VOClass.cpp:153-226, Source/VOClassUtils.cpp:20-59).
  image_0, image_1   left / right frames, 6-digit zero-padded ".png" names (Source/VOClass.cpp:36-83), read as
                     8-bit grey.
  estimated poses    "x y z" per line with the C++ stream default of 6 significant digits (Source/VOClass.cpp:613-615).
  depth .ply         ascii, one vertex "row col -z r g b" per pixel with z <= 3000 (Source/VOClassUtils.cpp:111-157,
                     call site Source/VOClass.cpp:346-360).

PNG decoding is done here (zlib + the five PNG row filters) for 8-bit grey, grey+alpha, RGB and RGBA non-interlaced
files, so that nothing outside the standard library and numpy is needed; colour is reduced with the fixed-point
weights cv::imread(IMREAD_GRAYSCALE) applies to 8-bit RGB (R 4899, G 9617, B 1868, >> 14, rounded).
"""
import os
import struct
import zlib

import numpy as np

from . import sequence

NAME_WIDTH = 6  # Source/VOClass.cpp:39


def image_name(frame_number):
    return "%0*d.png" % (NAME_WIDTH, frame_number)


# ------------------------------------------------------------------ calib / poses
def _numbers(line, skip):
    words = line.split()
    return [float(w) for w in words[skip:skip + 12]]


def read_calib(path):
    """projectionCL, projectionCR (3x4 float64) from the first two lines of calib.txt."""
    with open(path) as f:
        lines = [f.readline(), f.readline()]
    mats = []
    for line in lines:
        v = _numbers(line, 1)
        if len(v) != 12:
            raise ValueError("calibration line does not hold a label and 12 numbers: %r" % line)
        mats.append(np.array(v, dtype=np.float64).reshape(3, 4))
    return mats[0], mats[1]


def write_calib(path, P0, P1):
    with open(path, "w") as f:
        for name, P in (("P0:", P0), ("P1:", P1)):
            f.write(name + " " + " ".join("%.12e" % v for v in np.asarray(P, dtype=np.float64).reshape(12)) + "\n")


def read_poses(path):
    """All [R|t] lines of a poses file as n x 3 x 4; ground_truth() gives what the reference keeps of them."""
    out = []
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            v = _numbers(line, 0)
            if len(v) != 12:
                raise ValueError("pose line does not hold 12 numbers: %r" % line)
            out.append(np.array(v, dtype=np.float64).reshape(3, 4))
    return np.array(out).reshape(-1, 3, 4)


def ground_truth(poses):
    """The translation column of every pose: the reference's groundTruth vector."""
    return np.asarray(poses)[:, :, 3].copy()


def write_poses(path, poses):
    with open(path, "w") as f:
        for P in np.asarray(poses, dtype=np.float64).reshape(-1, 12):
            f.write(" ".join("%.9e" % v for v in P) + "\n")


def write_trajectory_xyz(path, trajectory):
    """One "x y z" line per pose, 6 significant digits like an unformatted C++ ostream."""
    with open(path, "w") as f:
        for p in np.asarray(trajectory, dtype=np.float64).reshape(-1, 3):
            f.write("%g %g %g\n" % (p[0], p[1], p[2]))


def read_trajectory_xyz(path):
    return np.loadtxt(path, dtype=np.float64).reshape(-1, 3)


# ------------------------------------------------------------------ PNG
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _unfilter(raw, h, stride, bpp):
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    pos = 0
    for y in range(h):
        ft = raw[pos]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=pos + 1).astype(np.int32)
        pos += stride + 1
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for c in range(bpp):  # running sum per byte lane, mod 256
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        elif ft in (3, 4):
            cur = np.zeros(stride, dtype=np.int32)
            lv, pv = line.tolist(), prev.tolist()
            cv = [0] * stride
            for i in range(stride):
                a = cv[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = pv[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cv[i] = (lv[i] + pred) & 255
            cur = np.array(cv, dtype=np.int32)
        else:
            raise ValueError("bad PNG filter type %d" % ft)
        out[y] = cur
        prev = cur
    return out


def _rgb_to_gray(px):
    """cv::imread(..., IMREAD_GRAYSCALE) of a colour file: fixed-point BT.601 weights, 14 fractional bits"""
    r, g, b = (px[:, :, i].astype(np.int32) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def read_png_gray(path, use_pil=True):
    """8-bit grey image of a PNG file (H x W uint8).  Decoded by Pillow when it is installed (a C decoder: ~7 ms for a
    KITTI frame); the built-in decoder below gives the same pixels and needs nothing but zlib, but its Average and Paeth
    rows are undone pixel by pixel in Python (~200 ms for such a frame)."""
    if use_pil:
        try:
            from PIL import Image
        except Exception:
            Image = None
        if Image is not None:
            with Image.open(path) as im:
                if im.format == "PNG" and im.mode in ("L", "RGB", "RGBA", "LA"):
                    a = np.array(im)
                    if a.dtype == np.uint8:
                        if a.ndim == 2:
                            return np.ascontiguousarray(a)
                        return np.ascontiguousarray(a[:, :, 0]) if a.shape[2] == 2 else _rgb_to_gray(a)
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _PNG_SIG:
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ValueError("no IHDR chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if depth != 8 or channels is None or interlace != 0:
        raise ValueError("unsupported PNG layout (bit depth %d, colour type %d, interlace %d)" % (depth, ctype, interlace))
    px = _unfilter(zlib.decompress(b"".join(idat)), h, w * channels, channels).reshape(h, w, channels)
    if channels <= 2:
        return np.ascontiguousarray(px[:, :, 0])
    return _rgb_to_gray(px)


def write_png_gray(path, img, filter_type=0):
    """Non-interlaced 8-bit grey PNG; filter_type 0..4 is applied to every row (all five exist so that the reader can be tested)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    rows = []
    prev = np.zeros(w, dtype=np.int32)
    for y in range(h):
        cur = img[y].astype(np.int32)
        left = np.concatenate([[0], cur[:-1]])
        ul = np.concatenate([[0], prev[:-1]])
        if filter_type == 0:
            enc = cur
        elif filter_type == 1:
            enc = cur - left
        elif filter_type == 2:
            enc = cur - prev
        elif filter_type == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = cur - pred
        rows.append(bytes([filter_type]) + (enc & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(b"".join(rows), 6)) + chunk(b"IEND", b""))


# ------------------------------------------------------------------ sequence directory
class Sequence:
    """<root>/image_0, <root>/image_1, <root>/calib.txt; optional poses file."""

    def __init__(self, root, poses_file=None):
        self.root = root
        self.left_dir = os.path.join(root, "image_0")
        self.right_dir = os.path.join(root, "image_1")
        self.P0, self.P1 = read_calib(os.path.join(root, "calib.txt"))
        self.poses = read_poses(poses_file) if poses_file else None
        n = 0
        while os.path.exists(os.path.join(self.left_dir, image_name(n))) and os.path.exists(os.path.join(self.right_dir, image_name(n))):
            n += 1
        self.n_frames = n

    def __len__(self):
        return self.n_frames

    def frame(self, i):
        return read_png_gray(os.path.join(self.left_dir, image_name(i))), read_png_gray(os.path.join(self.right_dir, image_name(i)))


def write_sequence(root, frames, P0, P1, poses=None, poses_file=None):
    """Writes frames [(left, right), ...] in the KITTI layout (used by the tests and to export synthetic sequences)."""
    os.makedirs(os.path.join(root, "image_0"), exist_ok=True)
    os.makedirs(os.path.join(root, "image_1"), exist_ok=True)
    for i, (l, r) in enumerate(frames):
        write_png_gray(os.path.join(root, "image_0", image_name(i)), l, filter_type=i % 5)
        write_png_gray(os.path.join(root, "image_1", image_name(i)), r, filter_type=(i + 2) % 5)
    write_calib(os.path.join(root, "calib.txt"), P0, P1)
    if poses is not None:
        write_poses(poses_file or os.path.join(root, "poses.txt"), poses)


def run(root, poses_file=None, out_file=None, batch=16, limit=None, device=0, decode_threads=16, **params):
    """The reference's main() (Source/main.cpp:7-74) on a KITTI sequence directory: returns (trajectory [n,3], RMSE or None)."""
    from . import abi
    seq = Sequence(root, poses_file)
    n = len(seq) if limit is None else min(len(seq), limit)
    if n < 2:
        raise ValueError("need at least two frames in %s" % root)
    first = seq.frame(0)[0]
    h, w = first.shape
    p = abi.default_params(w, h, max_batch=batch, **params)
    ctx = abi.Context(p, device)
    try:
        ctx.set_calib(seq.P0, seq.P1)
        Rts = []
        # chunks of `batch` pairs share their boundary frame.  Frames are decoded on worker threads (the PNG decoder
        # releases the interpreter lock), one chunk ahead of the chunk the GPU is working on.
        from concurrent.futures import ThreadPoolExecutor
        chunks = [(s, min(s + batch, n - 1)) for s in range(0, n - 1, batch)]
        with ThreadPoolExecutor(max_workers=max(1, min(decode_threads, os.cpu_count() or 1))) as pool:
            def request(k, have_first):
                s, e = chunks[k]
                return [pool.submit(seq.frame, i) for i in range(s + (1 if have_first else 0), e + 1)]
            pending = request(0, False)
            last = None
            for k in range(len(chunks)):
                frames = ([last] if last is not None else []) + [f.result() for f in pending]
                last = frames[-1]
                pending = request(k + 1, True) if k + 1 < len(chunks) else []
                Rt, _ = sequence.run_sequence(ctx, frames, batch)
                Rts.extend(Rt)
        traj, _ = sequence.integrate(Rts)
    finally:
        ctx.close()
    if out_file:
        write_trajectory_xyz(out_file, traj)
    err = None
    if seq.poses is not None:
        err = sequence.trajectory_rmse(traj, ground_truth(seq.poses)[:len(traj)])
    return traj, err


# ------------------------------------------------------------------ depth point cloud
def write_depth_ply(path, depth, gray, depth_thresh=3000):
    """VOClass::writeToPLY as called from computeDepthMap: vertices "row col -z" + grey as r g b for every pixel whose
    depth, narrowed to float, is <= depth_thresh.  The header's vertex count follows the reference's formula
    (pixels minus the histogram bin of the largest integer depth), which equals the number of lines written whenever
    the only depths above the threshold are the ones produced by the 0.1 disparity patch."""
    depth = np.asarray(depth, dtype=np.float64)
    gray = np.asarray(gray, dtype=np.uint8)
    z = depth.astype(np.float32)
    max_depth = float(depth.max())
    n_vertices = depth.size - int(np.count_nonzero(depth.astype(np.int64) == int(max_depth)))
    rows, cols = np.nonzero(~(z > depth_thresh))
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n_vertices)
        zz = -z[rows, cols]
        gg = gray[rows, cols]
        f.write("".join("%d %d %g %d %d %d\n" % (r, c, v, g, g, g) for r, c, v, g in zip(rows.tolist(), cols.tolist(), zz.tolist(), gg.tolist())))
    return len(rows), n_vertices


def write_cloud_ply(path, cloud, n_vertices=None):
    """The same file from a cloud compacted on the GPU (Context.depth_cloud): one "row col -z g g g" line per record.
    n_vertices: the header's count (the reference computes it with its own formula, see write_depth_ply); default: the
    number of records."""
    n_vertices = len(cloud) if n_vertices is None else n_vertices
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n_vertices)
        f.write("".join("%d %d %g %d %d %d\n" % (r, c, v, g, g, g)
                        for r, c, v, g in zip(cloud["row"].tolist(), cloud["col"].tolist(), cloud["neg_z"].tolist(), cloud["gray"].tolist())))
    return len(cloud)


def main(argv=None):
    """python -m visualodometry_amd.kitti SEQUENCE_DIR [--poses FILE] [--out FILE] [--batch B] [--limit N] [--device D]
    The reference's main() on a KITTI odometry sequence directory (image_0/, image_1/, calib.txt): writes the "x y z"
    pose dump the reference writes to Log/outputPoses.txt and prints the translation RMSE when ground truth is given."""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m visualodometry_amd.kitti", description=main.__doc__)
    ap.add_argument("sequence_dir")
    ap.add_argument("--poses", default=None, help="ground-truth poses file (12 numbers per line)")
    ap.add_argument("--out", default="outputPoses.txt")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--limit", type=int, default=None, help="use only the first N frames")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    traj, err = run(a.sequence_dir, poses_file=a.poses, out_file=a.out, batch=a.batch, limit=a.limit, device=a.device)
    print("%d poses written to %s" % (len(traj), a.out))
    if err is not None:
        print("Measured error: %g" % err)   # as the reference prints it (Source/main.cpp:63-65)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
