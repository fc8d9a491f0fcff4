"""N3: the reference's dataset layout (trajectory.txt + img/*.png + depth/*.depth), CPU only."""
import os
import struct
import zlib

import numpy as np

from rpg_svo_amd import dataset, se3, synth


def test_png_roundtrip_and_filters(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    dataset.write_png_gray8(p, img)
    assert np.array_equal(dataset.read_png_gray8(p), img)
    # a PNG written with every filter type (as real encoders do) decodes to the same pixels
    h, w = img.shape
    raw = bytearray()
    prev = np.zeros(w, dtype=np.int32)
    for y in range(h):
        line = img[y].astype(np.int32)
        ft = y % 5
        left = np.concatenate([[0], line[:-1]])
        ul = np.concatenate([[0], prev[:-1]])
        if ft == 0:
            enc = line
        elif ft == 1:
            enc = line - left
        elif ft == 2:
            enc = line - prev
        elif ft == 3:
            enc = line - ((left + prev) >> 1)
        else:
            pp = left + prev - ul
            pa, pb, pc = abs(pp - left), abs(pp - prev), abs(pp - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = line - pred
        raw += bytes([ft]) + (enc & 255).astype(np.uint8).tobytes()
        prev = line

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    q = str(tmp_path / "b.png")
    with open(q, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    assert np.array_equal(dataset.read_png_gray8(q), img)


def test_dataset_roundtrip(tmp_path):
    cam = synth.Camera(160, 120, 100.0, 100.0, 80.0, 60.0)
    T = synth.make_trajectory(4, seed=1)
    imgs = synth.render(synth.make_texture(seed=12345), T, cam).numpy()
    # z-depth of the plane z=0 seen from frame 0
    R = T[0, :9].reshape(3, 3); c = -R.T @ T[0, 9:]
    u, v = np.meshgrid(np.arange(cam.width, dtype=np.float64), np.arange(cam.height, dtype=np.float64))
    d = np.stack([(u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u)], -1)
    dw = d @ R
    z = (-c[2] / dw[..., 2]).astype(np.float32)  # ray parameter with d_z = 1 == z-depth
    root = str(tmp_path / "ds")
    names = dataset.write_dataset(root, imgs, T, cam, z_depth={0: z})
    ts, names2, T2 = dataset.read_trajectory_file(root)
    assert names2 == names and np.allclose(ts, np.arange(4) / 30.0)
    assert se3.log_norm(T2, T).max() < 1e-7
    for i, n in enumerate(names):
        assert np.array_equal(dataset.read_image(root, n), imgs[i])
    rng_map = dataset.load_blender_depthmap(os.path.join(root, "depth", names[0] + "_0.depth"), cam)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "dropin"))
    import pypipeline as pp
    assert np.allclose(rng_map, pp.range_map(cam, T[0]), rtol=2e-5)
