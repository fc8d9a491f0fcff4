"""Reader / writer of the dataset layout the reference's benchmark and tests consume (SURVEY 8f N3;
svo_ros/src/benchmark_node.cpp:178-256, svo/test/test_sparse_img_align.cpp:57-75, the
`sin2_tex2_h1_v8_d` Blender datasets):

    <dir>/trajectory.txt          one line per frame: timestamp image_name tx ty tz qx qy qz qw  (T_w_f;
                                  vk::blender_utils::file_format::ImageNameAndPose, '#' comments)
    <dir>/img/<image_name>_0.png  8-bit grayscale image
    <dir>/depth/<image_name>_0.depth   whitespace-separated z-depths, row-major, one per pixel
                                  (vk::blender_utils::loadBlenderDepthmap turns z into range along the ray)

No imaging library is available offline, so 8-bit grayscale PNG is (de)coded here with zlib.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

from . import se3, trace


# ---- minimal PNG (8-bit grayscale, non-interlaced) ------------------------------------------------
def write_png_gray8(path: str, img: np.ndarray) -> None:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    raw = np.zeros((h, w + 1), dtype=np.uint8)  # filter type 0 in front of every row
    raw[:, 1:] = img

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def read_png_gray8(path: str) -> np.ndarray:
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError(f"{path}: only 8-bit non-interlaced PNG is supported (depth {depth}, colour type {ctype})")
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:  # up
            cur = (line + prev) & 255
        elif ft == 1:  # sub
            cur = line.copy()
            for c in range(ch):
                cur[c::ch] = np.cumsum(line[c::ch]) & 255
        else:          # average / paeth: byte-serial
            cur = np.zeros_like(line)
            for i in range(w * ch):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    if ch == 1:
        return out
    px = out.reshape(h, w, ch)
    if ch == 2:
        return px[..., 0].copy()
    rgb = px[..., :3].astype(np.float32)  # cv::imread(..., 0): BT.601 luma
    return np.clip(np.round(0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]), 0, 255).astype(np.uint8)


# ---- depth maps ---------------------------------------------------------------------------------------
def write_depth_z(path: str, z: np.ndarray) -> None:
    with open(path, "w") as f:
        for row in np.asarray(z, dtype=np.float32):
            f.write(" ".join("%.6f" % v for v in row) + "\n")


def load_blender_depthmap(path: str, cam) -> np.ndarray:
    """vk::blender_utils::loadBlenderDepthmap: z-depth file -> range along each pixel's ray, float32 [h,w]."""
    z = np.loadtxt(path, dtype=np.float32).reshape(cam.height, cam.width)
    u, v = np.meshgrid(np.arange(cam.width, dtype=np.float64), np.arange(cam.height, dtype=np.float64))
    x, y = (u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy
    return (z * np.sqrt(x * x + y * y + 1.0)).astype(np.float32)


# ---- whole datasets -----------------------------------------------------------------------------------
def write_dataset(root: str, images: np.ndarray, T_f_w: np.ndarray, cam, timestamps=None, z_depth=None) -> list[str]:
    """images [n,h,w] u8, T_f_w [n,12]; z_depth: optional [n,h,w] (or {index: [h,w]}) z-depth maps."""
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    os.makedirs(os.path.join(root, "depth"), exist_ok=True)
    n = len(images)
    ts = np.arange(n) / 30.0 if timestamps is None else np.asarray(timestamps)
    names = ["frame_%06d" % i for i in range(n)]
    T_w_f = se3.inv(np.asarray(T_f_w, dtype=np.float64))
    with open(os.path.join(root, "trajectory.txt"), "w") as f:
        f.write("# timestamp image_name tx ty tz qx qy qz qw\n")
        for i in range(n):
            q = trace.quat_from_R(T_w_f[i, :9].reshape(3, 3))
            p = T_w_f[i, 9:]
            f.write("%.9f %s %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (ts[i], names[i], p[0], p[1], p[2], q[0], q[1], q[2], q[3]))
            write_png_gray8(os.path.join(root, "img", names[i] + "_0.png"), images[i])
    if z_depth is not None:
        items = z_depth.items() if isinstance(z_depth, dict) else enumerate(z_depth)
        for i, z in items:
            write_depth_z(os.path.join(root, "depth", names[i] + "_0.depth"), z)
    return names


def read_trajectory_file(root: str):
    """-> (timestamps [n], names [n], T_f_w [n,12])"""
    ts, names, T = [], [], []
    for line in open(os.path.join(root, "trajectory.txt")):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        tok = line.split()
        ts.append(float(tok[0]))
        names.append(tok[1])
        v = [float(x) for x in tok[2:9]]
        T_w_f = np.concatenate([trace.R_from_quat(np.array(v[3:7])).reshape(9), v[0:3]])
        T.append(se3.inv(T_w_f[None])[0])
    return np.array(ts), names, np.array(T)


def read_image(root: str, name: str) -> np.ndarray:
    return read_png_gray8(os.path.join(root, "img", name + "_0.png"))
