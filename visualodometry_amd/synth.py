"""Seeded synthetic stereo sequences with exact geometry (test and benchmark input, no dataset needed).

A straight textured corridor (ground plane, two side walls, a far backdrop) is rendered by per-pixel
ray/plane intersection for a rectified stereo rig with KITTI-00 intrinsics; the camera advances
0.8 m per frame with a zero-mean sinusoidal yaw.  Textures are sums of octaves of integer-lattice value
noise hashed from (seed, plane, octave, ix, iy), so images are identical on every machine.  Ground-truth
poses use the KITTI 12-number row-major [R|t] convention that the reference parses at
Source/VOClass.cpp:140-152, and the calibration the P0/P1 convention of Source/VOClass.cpp:85-101.
"""
import math
import numpy as np

KITTI_W, KITTI_H = 1241, 376
KITTI_FX, KITTI_CX, KITTI_CY = 718.856, 607.1928, 185.2157
KITTI_P1_03 = -386.1448
# KITTI odometry sequence lengths 00..10
KITTI_SEQ_LENGTHS = (4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201)


def calib(width=KITTI_W, height=KITTI_H, baseline_scale=1.0):
    """P0, P1 (3x4 f64).  Other sizes scale the KITTI intrinsics by width/1241 and keep the 0.537 m baseline
    (times baseline_scale)."""
    s = width / float(KITTI_W)
    fx = KITTI_FX * s
    cx = KITTI_CX * s
    cy = KITTI_CY * (height / float(KITTI_H))
    b = -KITTI_P1_03 / KITTI_FX * baseline_scale
    P0 = np.array([[fx, 0, cx, 0], [0, fx, cy, 0], [0, 0, 1, 0]], dtype=np.float64)
    P1 = P0.copy()
    P1[0, 3] = -fx * b
    return P0, P1


def pixel_matched_scene(width):
    """Sequence(...) keyword arguments that give an image `width` pixels wide the per-pixel statistics of the 1241-pixel
    one: texture wavelengths, stereo baseline, forward step and yaw rate all scaled by 1241 / width, so that corner density
    per pixel, disparities and frame-to-frame flow (in pixels) are those of the KITTI-shape sequence.  Without it a wider
    image of the same scene shows the same surfaces magnified: fewer corners per pixel and flows beyond the reach of a
    4-level pyramid (at 3840 pixels only a third of the features close the matching circle)."""
    s = KITTI_W / float(width)
    return dict(step=0.8 * s, yaw_rate=0.02 * s, texture_scale=s, baseline_scale=s)


def _hash2(seed, plane, octave, ix, iy):
    """Integer lattice hash -> uint32 array (wrap-around uint64 arithmetic, identical everywhere)."""
    with np.errstate(over="ignore"):
        h = (ix.astype(np.int64).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + iy.astype(np.int64).astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
             + np.uint64(seed) * np.uint64(0x165667B19E3779F9)
             + np.uint64(plane * 131 + octave + 1) * np.uint64(0x27D4EB2F165667C5))
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(29)
    return (h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def _value_noise(seed, plane, octave, a, b):
    """Bilinear value noise on the unit lattice; a, b f64 arrays."""
    ia = np.floor(a)
    ib = np.floor(b)
    fa = a - ia
    fb = b - ib
    ia = ia.astype(np.int64)
    ib = ib.astype(np.int64)
    v00 = _hash2(seed, plane, octave, ia, ib)
    v10 = _hash2(seed, plane, octave, ia + 1, ib)
    v01 = _hash2(seed, plane, octave, ia, ib + 1)
    v11 = _hash2(seed, plane, octave, ia + 1, ib + 1)
    return (v00 * (1 - fa) + v10 * fa) * (1 - fb) + (v01 * (1 - fa) + v11 * fa) * fb


# lattice spacing (m) and amplitude per octave; a blocky (piecewise constant) layer adds hard corners
_OCTAVES = ((1.6, 0.6), (0.8, 0.6), (0.4, 0.6), (0.2, 0.45), (0.1, 0.3))
_BLOCK = (0.6, 0.5)


def _texture(seed, plane, a, b, footprint):
    """Texture value (roughly zero mean) at plane coordinates (a, b) in metres.
    footprint = metres per pixel at the hit point; octaves finer than ~2 px fade out (LOD)."""
    acc = np.zeros_like(a)
    for o, (spacing, amp) in enumerate(_OCTAVES):
        lod = np.clip(spacing / (2.0 * footprint) - 0.5, 0.0, 1.0)
        acc += amp * lod * (_value_noise(seed, plane, o, a / spacing, b / spacing) - 0.5)
    spacing, amp = _BLOCK
    lod = np.clip(spacing / (4.0 * footprint) - 0.5, 0.0, 1.0)
    blk = _hash2(seed, plane, 77, np.floor(a / spacing).astype(np.int64), np.floor(b / spacing).astype(np.int64))
    acc += amp * lod * (blk - 0.5)
    return acc


class Sequence:
    """One synthetic stereo sequence.  pose(i) is the 3x4 camera-to-world [R|t] of the left camera."""

    def __init__(self, seed=1000, width=KITTI_W, height=KITTI_H, step=0.8, yaw_rate=0.02, yaw_period=50.0,
                 wall=7.0, ground=1.65, far=6000.0, contrast=130.0, texture_scale=1.0, baseline_scale=1.0):
        """texture_scale: every texture wavelength (in metres on the surfaces) is multiplied by it; baseline_scale: the
        stereo baseline is.  With both, `step` and `yaw_rate` set to 1241 / width, an image of `width` pixels has the texture
        statistics, disparities and frame-to-frame flow, all counted in pixels, of the 1241-pixel one."""
        self.seed = int(seed)
        self.width, self.height = int(width), int(height)
        self.texture_scale = float(texture_scale)
        self.P0, self.P1 = calib(width, height, baseline_scale)
        self.fx, self.fy = self.P0[0, 0], self.P0[1, 1]
        self.cx, self.cy = self.P0[0, 2], self.P0[1, 2]
        self.baseline = self.P1[0, 3] / (-self.fx)
        self.step, self.yaw_rate, self.yaw_period = step, yaw_rate, yaw_period
        self.wall, self.ground, self.far = wall, ground, far
        self.contrast = float(contrast)
        self._poses = [(0.0, 0.0, 0.0)]  # (x, z, yaw)

    def _state(self, i):
        while len(self._poses) <= i:
            k = len(self._poses) - 1
            x, z, yaw = self._poses[-1]
            x += self.step * math.sin(yaw)
            z += self.step * math.cos(yaw)
            yaw += self.yaw_rate * math.cos(2.0 * math.pi * k / self.yaw_period)
            self._poses.append((x, z, yaw))
        return self._poses[i]

    def pose(self, i):
        x, z, yaw = self._state(i)
        c, s = math.cos(yaw), math.sin(yaw)
        return np.array([[c, 0, s, x], [0, 1, 0, 0], [-s, 0, c, z]], dtype=np.float64)

    def poses_kitti(self, n):
        """n x 12 ground-truth rows (KITTI poses/XX.txt layout), relative to frame 0 (identity)."""
        return np.stack([self.pose(i).reshape(12) for i in range(n)])

    def render(self, i, right=False):
        """u8 H x W image of frame i for the left (or right) camera."""
        T = self.pose(i)
        R, t = T[:, :3], T[:, 3]
        o = t + (R[:, 0] * self.baseline if right else 0.0)
        u = np.arange(self.width, dtype=np.float64)
        v = np.arange(self.height, dtype=np.float64)
        dx = (u - self.cx) / self.fx
        dy = (v - self.cy) / self.fy
        DX, DY = np.meshgrid(dx, dy)
        dirx = R[0, 0] * DX + R[0, 1] * DY + R[0, 2]
        diry = R[1, 0] * DX + R[1, 1] * DY + R[1, 2]
        dirz = R[2, 0] * DX + R[2, 1] * DY + R[2, 2]
        big = 1e30
        with np.errstate(divide="ignore", invalid="ignore"):
            sg = np.where(diry > 1e-12, (self.ground - o[1]) / diry, big)
            swl = np.where(dirx < -1e-12, (-self.wall - o[0]) / dirx, big)
            swr = np.where(dirx > 1e-12, (self.wall - o[0]) / dirx, big)
            sb = np.where(dirz > 1e-12, (self.far - o[2]) / dirz, big)
        s = np.minimum(np.minimum(sg, sb), np.minimum(swl, swr))
        plane = np.where(s == sg, 0, np.where(s == swl, 1, np.where(s == swr, 2, 3)))
        px = o[0] + s * dirx
        py = o[1] + s * diry
        pz = o[2] + s * dirz
        footprint = s / self.fx  # s is depth along the optical axis because dir_cam.z == 1
        img = np.zeros((self.height, self.width), dtype=np.float64)
        for pid, (a, b) in enumerate(((px, pz), (pz, py), (pz, py), (px, py))):
            m = plane == pid
            if not m.any():
                continue
            scale = (1.0 if pid < 3 else 40.0) * self.texture_scale  # the backdrop is far away: use a coarser texture there
            img[m] = _texture(self.seed, pid, a[m] / scale, b[m] / scale, footprint[m] / scale)
        out = np.rint(img * self.contrast + 128.0)
        return np.clip(out, 0, 255).astype(np.uint8)

    def frame(self, i):
        """(left, right) u8 images of frame i."""
        return self.render(i, False), self.render(i, True)


def random_image(seed, width, height, smooth=2):
    """Small seeded test image: random blocks + noise, deterministic."""
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=((height + 7) // 8 + 1, (width + 7) // 8 + 1)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8)))[:height, :width]
    img += rng.randint(-12, 13, size=img.shape)
    for _ in range(smooth):
        img[1:-1, 1:-1] = (img[:-2, 1:-1] + img[2:, 1:-1] + img[1:-1, :-2] + img[1:-1, 2:] + 4 * img[1:-1, 1:-1]) / 8.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
